"""CAGroup3D detector (mirror of pcdet/models/detectors/cagroup3d.py:8-157): colour normalisation,
voxelisation into a sparse tensor (hash build), backbone -> dense head -> RoI head, loss / prediction
packaging.  Same forward(batch_dict) contract as the reference."""
import torch

from .... import me as ME
from ..dense_heads.cagroup_head import split_gt_boxes
from .detector3d_template import Detector3DTemplate


class CAGroup3D(Detector3DTemplate):
    def __init__(self, model_cfg, num_class, dataset):
        super().__init__(model_cfg=model_cfg, num_class=num_class, dataset=dataset)
        self.module_list = self.build_networks()
        self.voxel_size = self.model_cfg.VOXEL_SIZE
        self.semantic_min_threshold = self.model_cfg.SEMANTIC_MIN_THR
        self.semantic_iter_value = self.model_cfg.SEMANTIC_ITER_VALUE
        self.semantic_value = self.model_cfg.SEMANTIC_THR

    def voxelization(self, points):
        """points (N,7) = (b,x,y,z,r,g,b) -> sparse tensor on the 0.02 m grid; one (the first) point's
        colour per voxel (cagroup3d.py:18-25)."""
        coordinates = points[:, :4].clone()
        coordinates[:, 1:] /= self.voxel_size
        return ME.SparseTensor(coordinates=coordinates, features=points[:, 4:].clone())

    def forward(self, batch_dict):
        cur_epoch = batch_dict.get("cur_epoch", None)
        assert cur_epoch is not None
        self.module_list[1].semantic_threshold = max(self.semantic_value - int(cur_epoch) * self.semantic_iter_value,
                                                     self.semantic_min_threshold)
        batch_dict["points"][:, -3:] = batch_dict["points"][:, -3:] / 255.
        batch_dict["sp_tensor"] = self.voxelization(batch_dict["points"])
        for module in self.module_list:
            batch_dict.update(module(batch_dict))
        if self.training:
            loss, tb_dict, disp_dict = self.get_training_loss(batch_dict)
            disp_dict["cur_semantic_value"] = self.module_list[1].semantic_threshold
            return {"loss": loss}, tb_dict, disp_dict
        return self.post_processing(batch_dict)

    def post_processing(self, batch_dict):
        pred_dicts = [{"pred_boxes": batch_dict["batch_box_preds"][i], "pred_scores": batch_dict["batch_score_preds"][i],
                       "pred_labels": batch_dict["batch_cls_preds"][i]} for i in range(batch_dict["batch_size"])]
        recall_dict = {}
        if "gt_boxes" in batch_dict:
            recall_dict = {"gt": 0}
            for t in self.model_cfg.POST_PROCESSING.RECALL_THRESH_LIST:
                recall_dict["roi_%s" % str(t)] = 0
                recall_dict["rcnn_%s" % str(t)] = 0
        return pred_dicts, recall_dict

    @staticmethod
    def convert2list(points, batch_size=None):
        if batch_size is None:
            batch_size = int(points[:, 0].max().int()) + 1
        return [points[points[:, 0] == i, 1:] for i in range(batch_size)]

    def get_training_loss(self, batch_dict):
        bs = batch_dict["batch_size"]
        dev = batch_dict["points"].device

        def masks(key):
            if key not in batch_dict:
                return None
            return [x.to(dev) if torch.is_tensor(x) else torch.from_numpy(x).to(dev) for x in batch_dict[key]]
        if batch_dict.get("gt_bboxes_3d", None) is None:
            batch_dict["gt_bboxes_3d"], batch_dict["gt_labels_3d"] = split_gt_boxes(batch_dict["gt_boxes"], torch.long)
        x, semantic_scores, voxel_offset = batch_dict["one_stage_results"]
        centernesses, bbox_preds, cls_scores, voxel_points = x
        loss_one, tb_dict = self.dense_head.loss(
            centernesses, bbox_preds, cls_scores, voxel_points, semantic_scores, voxel_offset,
            batch_dict["gt_bboxes_3d"], batch_dict["gt_labels_3d"], self.convert2list(batch_dict["points"], bs),
            [None] * bs, masks("semantic_mask"), masks("instance_mask"))
        loss_two, tb_two = self.roi_head.loss(batch_dict)
        tb_dict.update(tb_two)
        disp_dict = dict(tb_dict)
        loss_all = loss_one + loss_two
        tb_dict = {"loss_all": tb_dict["one_stage_loss"] + tb_dict["loss_two_stage"], **tb_dict}
        return loss_all, tb_dict, disp_dict
