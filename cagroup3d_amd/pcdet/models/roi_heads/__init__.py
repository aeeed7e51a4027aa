from .cagroup_roi_head import CAGroup3DRoIHead

__all__ = {"CAGroup3DRoIHead": CAGroup3DRoIHead}
