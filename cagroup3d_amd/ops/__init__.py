"""Op-level mirror of the reference's pcdet/ops packages used by CAGroup3D
(iou3d_nms, knn, rotated_iou), bound to the C-ABI in include/cagroup3d_hip.h."""
