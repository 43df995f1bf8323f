"""The decoder's caller-side glue: what DyanmicQueryTransformer.forward
does immediately before and after ``self.decoder(...)`` and what validate_3d does with the result --
without the reference's CUDA-only constructor (lib/models/dq_transformer.py:120-205).

  person_joint_queries          dq_transformer.py:394-432  (query_embed_type 'person_joint')
  sample_space_reference_points dq_transformer.py:298-323 + generate_T_pose :225-236
  inverse_sigmoid               lib/models/util/misc.py:608-612
  decoder_outputs_to_dict       dq_transformer.py:569-603 (incl. the Shelf/Campus joint permutation)
  pack_predictions              lib/core/function.py:386-396 ([x, y, z, (score > thr) - 1, score])
  DecoderHead                   embeddings + decoder behind one forward(src_views, meta, threshold)
"""
from __future__ import annotations

import math

import torch
from torch import nn

from .synthetic import TPOSE_MM


def inverse_sigmoid(x, eps=1e-5):
    x = x.clamp(min=0, max=1)
    x1 = x.clamp(min=eps)
    x2 = (1 - x).clamp(min=eps)
    return torch.log(x1 / x2)


def person_joint_queries(joint_embedding_weight, instance_embedding_weight, batch):
    """(J, 2C) + (NQ, 2C) -> query_pos, tgt, each (batch, NQ*J, C); token order q = i*J + j."""
    c = joint_embedding_weight.shape[1] // 2
    query_embeds = (joint_embedding_weight.unsqueeze(0) + instance_embedding_weight.unsqueeze(1)).flatten(0, 1)
    query_embed, tgt = torch.split(query_embeds, c, dim=1)
    return query_embed.unsqueeze(0).expand(batch, -1, -1), tgt.unsqueeze(0).expand(batch, -1, -1)


def sample_space_reference_points(num_instance, space_size, space_center, batch, device, t_pose=None):
    """'sample_space' initial 3D query poses: ceil(sqrt(NQ))^2 xy grid at mid height, first NQ cells,
    norm2absolute, + T-pose joint offsets -> (batch, NQ*J, 3) float32 mm."""
    N = math.ceil(pow(num_instance, 1 / 2.0))
    x_ = torch.linspace(0., 1., N, device=device)
    z_ = torch.zeros(N, N, device=device) + 0.5
    x, y = torch.meshgrid(x_, x_, indexing="ij")
    root = torch.cat([x.unsqueeze(-1), y.unsqueeze(-1), z_.unsqueeze(-1)], dim=-1).view(-1, 3)[:num_instance]
    size = torch.as_tensor(space_size, dtype=torch.float32, device=device)
    center = torch.as_tensor(space_center, dtype=torch.float32, device=device)
    root_abs = root * size + center - size / 2.0
    if t_pose is None:
        t_pose = torch.from_numpy(TPOSE_MM)
    joints = root_abs.unsqueeze(1) + t_pose.to(device)                 # float64 T-pose promotes, as in the reference
    return joints.expand(batch, -1, -1, -1).reshape(batch, -1, 3).float()


def level_tables(src_views):
    """(L, 2) long spatial shapes and (L,) level starts from the backbone's maps (dq_transformer.py:360-388)."""
    dev = src_views[0].device
    shapes = torch.as_tensor([list(s.shape[-2:]) for s in src_views], dtype=torch.long, device=dev)
    starts = torch.cat([shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]])
    return shapes, starts


def decoder_outputs_to_dict(hs, inter_references, inter_references_2d, inter_references_2d_projs, outputs_classes,
                            num_instance, num_joints, convert_joint_format_indices=None):
    """final-layer ``out`` dict of DyanmicQueryTransformer.forward (+ per-layer lists)."""
    batch = hs.shape[1]
    logits = [inverse_sigmoid(c) for c in outputs_classes]
    coords, coords2d, coords2dp = [], [], []
    for lvl in range(hs.shape[0]):
        c3, c2, cp = inter_references[lvl], inter_references_2d[lvl], inter_references_2d_projs[lvl]
        if convert_joint_format_indices is not None:
            idx = list(convert_joint_format_indices)
            c3 = c3.view(batch, num_instance, num_joints, -1)[..., idx, :].flatten(1, 2)
            nv = c2.shape[1]
            c2 = c2.view(batch, nv, num_instance, num_joints, -1)[..., idx, :].flatten(2, 3)
            cp = cp.view(batch, nv, num_instance, num_joints, -1)[..., idx, :].flatten(2, 3)
        coords.append({"outputs_coord": c3})
        coords2d.append({"outputs_coord_2d": c2})
        coords2dp.append({"outputs_coord_2d_proj": cp})
    return {"pred_logits": logits[-1], "pred_poses": coords[-1], "pred_poses_2d": coords2d[-1],
            "pred_poses_2d_proj": coords2dp[-1], "all_logits": logits, "all_poses": coords}


def pack_predictions(out, threshold):
    """(B, NQ, J', 5) = [x, y, z, (score > thr) - 1, score], the array validate_3d hands to NMS / evaluation."""
    logits = out["pred_logits"]
    bs, nq = logits.shape[:2]
    poses = out["pred_poses"]["outputs_coord"]
    nj = poses.shape[1] // nq
    poses = poses.view(bs, nq, nj, 3)
    score = logits[:, :, 1:2].sigmoid().unsqueeze(2).expand(-1, -1, nj, -1)
    return torch.cat([poses, (score > threshold).float() - 1, score], dim=-1)


class DecoderHead(nn.Module):
    """joint / instance embeddings + DQDecoder: the part of DyanmicQueryTransformer that sits behind the
    backbone.  Parameter names match the reference model (``joint_embedding.weight``,
    ``instance_embedding.weight``, ``decoder.layers.{i}.*``), so ``load_state_dict(ckpt, strict=False)``
    fills it from a published checkpoint."""

    def __init__(self, decoder, num_instance, num_joints, d_model, space_size, space_center,
                 convert_joint_format_indices=None, t_pose=None):
        super().__init__()
        self.t_pose = t_pose                       # (J, 3) mm; None = the reference's tpose.pt values (TPOSE_MM)
        self.decoder = decoder
        self.num_instance, self.num_joints = num_instance, num_joints
        self.joint_embedding = nn.Embedding(num_joints, d_model * 2)
        self.instance_embedding = nn.Embedding(num_instance, d_model * 2)
        self.space_size, self.space_center = list(space_size), list(space_center)
        self.convert_joint_format_indices = convert_joint_format_indices

    @torch.no_grad()
    def forward(self, src_views, meta, spatial_shapes=None, level_start_index=None, threshold=0.1):
        """src_views: list of L (V*B, C, H_l, W_l) backbone maps, view-major.  Returns (out dict, pred array)."""
        dev = src_views[0].device
        V = len(meta)
        batch = src_views[0].shape[0] // V
        if spatial_shapes is None:
            spatial_shapes, level_start_index = level_tables(src_views)
        query_pos, tgt = person_joint_queries(self.joint_embedding.weight, self.instance_embedding.weight, batch)
        ref = sample_space_reference_points(self.num_instance, self.space_size, self.space_center, batch, dev,
                                            t_pose=self.t_pose)
        hs, refs, refs2d, projs2d, classes = self.decoder(
            tgt.contiguous(), ref, src_views, meta, spatial_shapes, level_start_index, None,
            query_pos=query_pos.contiguous(), threshold=threshold)
        out = decoder_outputs_to_dict(hs, refs, refs2d, projs2d, classes, self.num_instance, self.num_joints,
                                      self.convert_joint_format_indices)
        return out, pack_predictions(out, threshold)
