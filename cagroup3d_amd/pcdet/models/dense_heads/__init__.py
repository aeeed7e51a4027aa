from .cagroup_head import CAGroup3DHead

__all__ = {"CAGroup3DHead": CAGroup3DHead}
