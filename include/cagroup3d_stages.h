/*
 * cagroup3d_stages.h -- stage-level fused operators of the two heads (C-ABI, part of libcagroup3d_hip.so).
 *
 * The reference writes these stages as chains of tensor expressions (tens of launches each over a few hundred to a few
 * hundred thousand rows, with the host reads that size them); each entry point below is ONE pass over the same data with
 * the same arithmetic in the same operation order, so that integer outputs are bit-identical and fp32 outputs differ at most
 * by the last-bit differences of exp / log / sin / cos between two math libraries.  Conventions as in cagroup3d_hip.h: raw
 * device pointers, caller-owned outputs, status codes, no allocation, no synchronisation.  Citations are into /root/reference.
 */
#ifndef CAGROUP3D_STAGES_H
#define CAGROUP3D_STAGES_H

#include "cagroup3d_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------------------------
 * Second stage, training: RoI <-> ground-truth matching, target construction, grid coordinates, regression loss.
 *
 * Proposals arrive FLAT: boxes float32 [M,7] (x,y,z,dx,dy,dz,heading in the dense head's convention), scores float32 [M],
 * labels int64 [M], scene-major with roi_off int32 [nb+1].  The reference first pads them to [nb, rin, .] with zero rows
 * (CAGroup3DRoIHead.reoder_rois_for_refining, pcdet/models/roi_heads/cagroup_roi_head.py:328-362: heading negated :358) and
 * enlarges the sizes (forward_train :270-272); padded row i of scene b is proposal roi_off[b] + i when that is below
 * roi_off[b+1] and an all-zero RoI with label 0 and score 0 otherwise.  Both entry points below address that padded index
 * space without materialising it.  Ground truth: gt_boxes float32 [nb, gmax, gdim >= 8] zero-padded, class id in column 7
 * (the batch_dict tensor itself), n_gt int32 [nb] real boxes per scene (a prefix); the heading is negated on the fly
 * (mmdet3d -> pcdet, cagroup_proposal_target_layer.py:97).
 *
 * cg3d_roi_match (ProposalTargetLayer.get_max_iou_with_same_class, cagroup_proposal_target_layer.py:204-238, with
 *   boxes_iou3d_gpu, pcdet/ops/iou3d_nms/iou3d_nms_utils.py:59-79): for every padded RoI the largest 3D IoU -- rotated BEV
 *   overlap x height overlap / max(vol_a + vol_b - overlap, 1e-6) -- over the ground-truth boxes of ITS scene and ITS class,
 *   and that box's index inside the scene (ties: the lowest index); a RoI whose class has no box in the scene gets overlap 0
 *   and index 0.   max_ov float32 [nb*rin], assign int32 [nb*rin].
 *
 * cg3d_roi_targets (sample_rois_for_rcnn gathers :44-63, ProposalTargetLayer.forward :22-33, CAGroup3DRoIHead.assign_targets
 *   cagroup_roi_head.py:291-326, and the regression targets of get_box_reg_layer_loss :551-577 = CAGroupResidualCoder.
 *   encode_torch, pcdet/models/model_utils/cagroup_utils.py:101-136, against the RoI with its centre (and heading) zeroed):
 *   keep int32 [nb*rsel] = the sampled padded index of every output row (the draw itself stays on the host: it follows the
 *   reference's two host RNG streams).  code_size = columns of the regression target: 6 (no heading), 7 (heading
 *   difference) or 8 (cos, sin of the heading: encode_angle_by_sincos).  Outputs, one row per sampled RoI:
 *     o_rois [.,7], o_gt_src [.,7] (the matched box, pcdet heading), o_gt [.,7] (the same box in the RoI's canonical frame),
 *     o_gt_label float32, o_iou float32, o_score float32, o_label int64, o_reg_valid int64 (iou > reg_fg),
 *     o_cls_label float32 (1 above cls_fg, 0 below cls_bg, (iou - cls_bg) / cls_span between), o_reg_target [., code_size].
 * ---------------------------------------------------------------------------------------------------------------- */
int cg3d_roi_match(const float *boxes, const int64_t *labels, const int32_t *roi_off, int32_t nb, int32_t rin, float enlarge,
                   const float *gt_boxes, int32_t gmax, int32_t gdim, const int32_t *n_gt, float *max_ov, int32_t *assign,
                   cg3d_stream_t stream);
int cg3d_roi_targets(const float *boxes, const float *scores, const int64_t *labels, const int32_t *roi_off, int32_t nb,
                     int32_t rin, float enlarge, const float *gt_boxes, int32_t gmax, int32_t gdim, const float *max_ov,
                     const int32_t *assign, const int32_t *keep, int32_t rsel, int32_t code_size, float reg_fg, float cls_fg,
                     float cls_bg, float cls_span, float *o_rois, float *o_gt_src, float *o_gt, float *o_gt_label,
                     float *o_iou, float *o_score, int64_t *o_label, int64_t *o_reg_valid, float *o_cls_label,
                     float *o_reg_target, cg3d_stream_t stream);

/* cg3d_roi_grid_coords (CAGroup3DRoIHead.get_dense_grid_points / get_global_grid_points_of_roi / roi_grid_pool,
 *   cagroup_roi_head.py:199-261, and the quantisation of SimplePoolingLayer.forward :46-68): the grid^3 cell centres of every
 *   RoI -- ((i + 0.5) / grid * size - size / 2, rotated by the heading when with_yaw, + centre), index order (ix, iy, iz) --
 *   quantised to floor(p / voxel_size), clamped to [clamp_lo, clamp_hi] (as floats, then truncated) and multiplied by coord_key.
 *   rois float32 [n,7]; coords int32 [n * grid^3, 4] = (scene = row / rois_per_scene, x, y, z), RoI-major.  Duplicates are
 *   kept: cg3d_coord_map_build merges them (the reference: linearise + torch.unique :54-67). */
int cg3d_roi_grid_coords(const float *rois, int64_t n, int32_t rois_per_scene, int32_t grid, int32_t with_yaw, float voxel_size,
                         float clamp_lo, float clamp_hi, int32_t coord_key, int32_t *coords, cg3d_stream_t stream);

/* cg3d_roi_reg_loss_{fwd,bwd} (get_box_reg_layer_loss, cagroup_roi_head.py:551-590 with WeightedSmoothL1Loss,
 *   pcdet/utils/loss_utils.py:76-137): loss = weight / max(#valid, 1) * sum over valid rows and codes of
 *   smooth_l1((reg - target) * code_w; beta); a NaN target takes the prediction (zero difference).
 *   reg / target float32 [m, cs], valid int64 [m], code_w float32 [cs] (may be NULL) -> out float32 [2] = (loss, #valid).
 *   bwd: g float32 [1] the upstream gradient (device), dreg float32 [m, cs]. */
int cg3d_roi_reg_loss_fwd(const float *reg, const float *target, const int64_t *valid, const float *code_w, int64_t m,
                          int32_t cs, float beta, float weight, float *out, cg3d_stream_t stream);
int cg3d_roi_reg_loss_bwd(const float *reg, const float *target, const int64_t *valid, const float *code_w, int64_t m,
                          int32_t cs, float beta, float weight, const float *fwd_out, const float *g, float *dreg,
                          cg3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Dense head: rows of the class branches (CAGroup3DHead.forward, pcdet/models/dense_heads/cagroup_head.py:209-258).
 *
 * For every class c the reference selects the backbone voxels with sigmoid(semantic score) > threshold plus one pad voxel
 * per scene (:227-233), and builds the point set [voted positions ; original positions] of the selection (:234-252), which it
 * quantises at the class voxel size and at `expand` times it (:254-271).  All classes live in ONE coordinate space here
 * (batch index c * nbatch + scene): the selection of class c is rows [start_c, start_c + n_c) of a class-major list (selected
 * voxels ascending, pads last), and its fused rows are [start_c * (nvote + 1), ...): first the n_c * nvote vote rows
 * (voxel-major), then the n_c original rows.
 *
 *   hit uint8 [n, nc] row-major (voxel selected for class), coords int32 [n,4] (scene, x, y, z) of the backbone voxels.
 * cg3d_class_count: block_off int32 [(nc + 6) * cg3d_class_nblk(n)]: its first nc * nblk entries = for every (class, block of
 *   1024 voxels) the number of selected voxels of that class in earlier blocks (the rest is scratch); totals int32 [nc + 6] = selected voxels per class, then the minimum and
 *   the maximum of the three coordinate columns (the scene bounds of :209-212).  The caller reads totals[0..nc) to size the
 *   outputs of cg3d_class_rows (the reference: torch.nonzero).
 * cg3d_class_rows: with E = sum(totals) + nc * nbatch and pad_row int32 [nbatch] (the first voxel of every scene),
 *   offsets float32 [n, nvote * 3] (the predicted votes), voxel_size / ts of the backbone map, vs_tab float32 [nc,3]:
 *   src int32 [E * (nvote + 1)] = row of every fused row in the table [votes (n * nvote rows) ; originals (n rows)],
 *   fine / coarse int32 [E * (nvote + 1), 4] = (c * nbatch + scene, floor(p / vs_c)), (.., floor(p / (vs_c * expand)) * expand)
 *   with p = the original position coords * voxel_size, or that plus the vote clamped to the scene bounds
 *   [(min - ts) * voxel_size, (max + ts) * voxel_size].
 * ---------------------------------------------------------------------------------------------------------------- */
int32_t cg3d_class_nblk(int64_t n);
int cg3d_class_count(const uint8_t *hit, int64_t n, int32_t nc, const int32_t *coords, int32_t *block_off, int32_t *totals,
                     cg3d_stream_t stream);
int cg3d_class_rows(const uint8_t *hit, int64_t n, int32_t nc, int32_t nbatch, const int32_t *block_off, const int32_t *totals,
                    const int32_t *coords, const int32_t *pad_row, const float *offsets, int32_t nvote, float voxel_size,
                    int32_t ts, const float *vs_tab, int32_t expand, int32_t *src, int32_t *fine, int32_t *coarse,
                    cg3d_stream_t stream);

/* Row gather from a table held in two pieces, and its adjoint (cagroup_head.py:238-252: the features of the fused rows are
 * rows of [vote features ; backbone features]; the reference concatenates the two first).
 *   out[i,:] = idx[i] < na ? Fa[idx[i],:] : Fb[idx[i] - na,:];   dFa / dFb accumulate (+=, the caller zero-fills them). */
int cg3d_gather_rows2(const float *Fa, const float *Fb, int64_t na, const int32_t *idx, float *out, int64_t n, int32_t c,
                      cg3d_stream_t stream);
int cg3d_scatter_add_rows2(const float *dout, const int32_t *idx, float *dFa, float *dFb, int64_t na, int64_t n, int32_t c,
                           cg3d_stream_t stream);

/* counts[v] = number of i with ids[i * stride] == v, 0 <= v < m (values outside are ignored); ids int32 (is64 == 0) or int64;
 * counts int64 [m], zero-filled by the call.  (torch.bincount on an id column without its min / max scans.) */
int cg3d_count_ids(const void *ids, int64_t n, int32_t stride, int32_t is64, int32_t m, int64_t *counts, cg3d_stream_t stream);
/* The same histogram plus, in counts[m] (counts int64 [m + 1]), the number of violations of "ids is non-decreasing with values in
 * [0, m)": descents ids[i-1] > ids[i] and values outside the range.  Zero there means the rows of id v are the consecutive range
 * [sum(counts[:v]), sum(counts[:v+1])) -- what the per-scene row lists of a batch-major map are read from in ONE launch. */
int cg3d_count_sorted_ids(const void *ids, int64_t n, int32_t stride, int32_t is64, int32_t m, int64_t *counts, cg3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Dense head: proposals (CAGroup3DHead._get_bboxes_single / _nms, cagroup_head.py:579-624,747-797), all scenes and class
 * maps at once.  Rows of the merged class maps are sorted by segment s = class map * nbatch + scene.
 *
 * cg3d_prop_keys: keys int64 [n] = seg << 32 | ~bits(smax) (smax >= 0: the row's best score): ascending key order is
 *   (segment ascending, score descending), equal keys keep their row order under a stable sort -- the reference's
 *   per-map `topk(NMS_PRE)` (:590-594) becomes "the first min(size, NMS_PRE) rows of every segment" of ONE sort.
 * cg3d_prop_entries: `order` int64 [n] = row of every sorted position; seg_start int32 [nseg] = first sorted position of
 *   every segment, cand_off int32 [nseg + 1] = first candidate of every segment (candidate t of segment s is sorted position
 *   seg_start[s] + t - cand_off[s]; ncand = cand_off[nseg], known to the caller).  Every (candidate t, class i) with scores[row, i] > thr (:763-765) is an entry of the
 *   NMS problem p = (s % nbatch) * nc + i:  ekeys[slot] = p << 54 | ~bits(score) << 22 | (t * nc + i), slots in no particular
 *   order (an ascending sort of the keys is the order the reference reaches: problem, score descending, (t, i) ascending);
 *   counts int32 [nbatch * nc + 1] = entries per problem, then their total (zero-filled by the call).
 *   Limits: nbatch * nc < 512, ncand * nc < 2^22.
 * cg3d_prop_gather: for the sorted entry keys the decoded boxes (_bbox_pred_to_bbox, :654-703: ndim 6 = no heading, heading
 *   column 0; ndim 8 = the 'fcaf3d' parametrisation), the boxes handed to NMS (heading negated when ndim == 8, :770) and the
 *   scores.  e_boxes / nms_boxes float32 [total,7], e_score float32 [total].
 * ---------------------------------------------------------------------------------------------------------------- */
int cg3d_prop_keys(const int64_t *seg, const float *smax, int64_t n, int64_t *keys, cg3d_stream_t stream);
int cg3d_prop_entries(const int64_t *order, const int32_t *seg_start, const int32_t *cand_off, int32_t nseg, int32_t ncand,
                      int32_t nbatch, const float *scores, int32_t nc, float thr, int64_t *ekeys, int32_t *counts, cg3d_stream_t stream);
int cg3d_prop_gather(const int64_t *ekeys, int64_t total, const int64_t *order, const int32_t *seg_start,
                     const int32_t *cand_off, int32_t nseg, int32_t nc, const float *points, const float *bbox_pred,
                     int32_t ndim, const float *scores, float *e_boxes, float *nms_boxes, float *e_score, cg3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Differentiable rotated 3D IoU of aligned box pairs (the SUN RGB-D box losses: IoU3DLoss, pcdet/utils/iou3d_loss.py:14-30,
 * over cal_iou_3d, pcdet/ops/rotated_iou/oriented_iou_loss.py:86-109; box2corners_th :6-35; oriented_box_intersection_2d,
 * box_intersection_2d.py:13-184 with sort_vertices, cuda_op/sort_vert_kernel.cu:15-134).  The reference runs ~80 tensor
 * launches around its one native kernel and lets autograd differentiate them; here one launch each way:
 *   pred / target float32 [n,7] (x, y, z, dx, dy, dz, heading) -> iou float32 [n];
 *   bwd: g float32 [n] (upstream gradient of iou) -> dpred float32 [n,7] = g * d iou / d pred (the 24 candidate vertices, the
 *   polygon order and the shoelace sum are recomputed; the gradient follows the reference's graph: through the corner and
 *   edge-intersection coordinates of the polygon's vertices, the height overlap and the volume of `pred`; masks and the
 *   vertex order carry no gradient; targets receive none).
 * ---------------------------------------------------------------------------------------------------------------- */
int cg3d_rotated_iou3d_fwd(const float *pred, const float *target, int64_t n, float *iou, cg3d_stream_t stream);
int cg3d_rotated_iou3d_bwd(const float *pred, const float *target, int64_t n, const float *g, float *dpred, cg3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Vote targets of the dense head, ScanNet form (CAGroup3DHead.loss, cagroup_head.py:454-498): every backbone voxel takes the
 * instance of its nearest raw point (kNN, k = 1) and votes for the centre of the ground-truth box nearest to that instance's
 * bounding-box centre.  The reference loops over torch.unique(instance ids) per scene; scenes hold the same number of raw
 * points here (np), instance ids lie in [0, ni).
 *
 * cg3d_instance_centers: xyz float32 [nb, np, 3], ins / sem int64 [nb, np] (instance and semantic id of every raw point),
 *   gt_ctr float32 [nb, gmax, 3] with n_gt int32 [nb] valid rows, n_classes: for every (scene, instance) the bounding box of
 *   its points, the semantic id of its first point, and from them  centers float32 [nb, ni, 3] = the centre of the
 *   ground-truth box nearest (Euclidean, ties: the lower index) to the bounding-box centre when the instance is an object
 *   (semantic id < n_classes; :470-478), (-10000, -10000, -10000) for an instance that is present but no object (:466), zeros
 *   for an id no point carries.  ws: 8 * nb * ni int32 of scratch.
 * cg3d_vote_targets: vox_xyz float32 [n,3], vox_scene int64 [n], nearest int64 [n] (index of the voxel's nearest raw point in
 *   its scene) -> off_t float32 [n,3] = centre of the nearest point's instance - voxel position (0 where that is below -100,
 *   :488-494), off_m float32 [n] = 1 where all three components are real votes.
 * ---------------------------------------------------------------------------------------------------------------- */
int cg3d_instance_centers(const float *xyz, const int64_t *ins, const int64_t *sem, int32_t nb, int32_t np, int32_t ni,
                          const float *gt_ctr, int32_t gmax, const int32_t *n_gt, int32_t n_classes, float *centers, int32_t *ws,
                          cg3d_stream_t stream);
int cg3d_vote_targets(const float *vox_xyz, const int64_t *vox_scene, const int64_t *nearest, int64_t n, const int64_t *ins,
                      int32_t np, const float *centers, int32_t ni, float *off_t, float *off_m, cg3d_stream_t stream);

/* Positives loss of the class maps, yaw form (SUN RGB-D; the counterpart of cg3d_pos_loss in cagroup3d_hip.h): centerness BCE +
 * rotated IoU loss over the positive points (cagroup_head.py:532-546) with the 'fcaf3d' box decode of the eight predictions
 * (dx-, dx+, dy-, dy+, dz-, dz+, sin(2a) ln q, cos(2a) ln q) (:689-703) and cal_iou_3d.  Same arguments as cg3d_pos_loss_{fwd,bwd}
 * with bbox_pred float32 [N,8] and bbox_t float32 [N,tstride >= 7] (x,y,z,dx,dy,dz,heading);
 * partial float32 [cg3d_pos_loss_yaw_nblocks(npos)][2]; bwd writes dcenterness[r], dbbox_pred[r, 0:8] for r in pos. */
int32_t cg3d_pos_loss_yaw_nblocks(int64_t npos);
int cg3d_pos_loss_yaw_fwd(const float *centerness, const float *bbox_pred, const float *points, const float *ctr_t,
                          const float *bbox_t, int32_t tstride, const int64_t *scene, const float *n_pos,
                          const float *ctr_denorm, const int64_t *pos, int64_t npos, float wc, float wb, float eps,
                          float *partial, cg3d_stream_t stream);
int cg3d_pos_loss_yaw_bwd(const float *centerness, const float *bbox_pred, const float *points, const float *ctr_t,
                          const float *bbox_t, int32_t tstride, const int64_t *scene, const float *n_pos,
                          const float *ctr_denorm, const int64_t *pos, int64_t npos, float wc, float wb, float eps,
                          const float *gscale, float *dcenterness, float *dbbox_pred, cg3d_stream_t stream);

/* Outputs of the class branches' prediction layers (CAGroup3DHead.forward_single, cagroup_head.py:627-652, for all class maps at
 * once): row i belongs to class map c_i = coords[i,0] / nbatch (coords int32 [n,4] of the merged class map).
 *   bbox_pred[i, 0:6] = exp(reg[i, 0:6] * scale[c_i]), bbox_pred[i, 6:] = reg[i, 6:]          (Scale + exp, :640-645)
 *   points[i, :]      = coords[i, 1:4] * vs_tab[c_i, :]                                        (:647-650)
 *   cls[i, c_i]      += boost (in place; the bench's forced-selection aid, 0 = off)
 * reg float32 [n, nd] (nd = 6 or 8), scale float32 [nc], vs_tab float32 [nc,3].
 * bwd: dreg[i, 0:6] = dbbox[i, 0:6] * bbox_pred[i, 0:6] * scale[c_i], dreg[i, 6:] = dbbox[i, 6:];
 *      dscale float32 [nc] (zero-filled by the call) += sum_i sum_k dbbox[i,k] * bbox_pred[i,k] * reg[i,k], k < 6. */
int cg3d_head_outputs_fwd(const float *reg, int32_t nd, const int32_t *coords, int64_t n, int32_t nbatch, const float *scale,
                          const float *vs_tab, int32_t nc, float boost, float *cls, float *bbox_pred, float *points,
                          cg3d_stream_t stream);
int cg3d_head_outputs_bwd(const float *dbbox, const float *bbox_pred, const float *reg, int32_t nd, const int32_t *coords, int64_t n,
                          int32_t nbatch, const float *scale, int32_t nc, float *dreg, float *dscale, cg3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CAGROUP3D_STAGES_H */
