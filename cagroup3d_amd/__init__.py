"""cagroup3d_amd -- MI355X-native hot path of CAGroup3D (sparse-voxel detector) behind the
reference's pcdet module API.  See DESIGN.md."""
__version__ = "0.1.0"
