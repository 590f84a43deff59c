"""ORACLE -- TEST INFRASTRUCTURE.  `SamAutomaticMaskGenerator(sam).generate(image)` with EVERY reference default
(sam2image.py:71,118: 32 x 32 point grid = 1024 prompts x 3 candidates, pred_iou_thresh 0.88, stability_score_thresh 0.95,
stability_score_offset 1.0, box_nms_thresh 0.7) on the frozen ViT-H embedding of tests/golden/sam_vit_h_full.npz, and the
`show_anns` id map of its records (sam2image.py:92-115): tests/golden/amg_vith_full.npz.

    python -m oracle.make_golden_amg_full            # ~6 minutes on 8 cores, ~6 GB of host memory
    python -m oracle.make_golden_amg_full --explore  # low-resolution statistics only (how the constants below were chosen)

This is the setting bench.py times as `with_amg` (ViT-H, the full grid, NMS on); the other AMG tests run a tiny encoder, 16
prompts and wide-open thresholds.  Nothing here is opened: the two score filters and the box NMS -- the only places where a
1e-6 move of a logit changes WHICH records exist, and with it every id above them -- all do real work (counts in the npz:
`n_candidates` -> `n_pass_iou` -> `n_pass_stability` -> `n_records`).

Random weights do not give a decoder whose numbers land near those thresholds (predicted IoUs around 0, logits of ~0.03,
masks that are all-ones or all-zeros), so the synthetic decoder is CALIBRATED by a state-dict edit that every consumer of the
golden repeats (`calibrated_decoder_state_dict`, a pure function of the seed):
  * one channel of the upscaled embedding becomes a constant ("bias channel": its transposed-convolution weights 0, bias 1),
    so the hypernetwork output for that channel is a per-mask offset of the logits; the offsets of the three candidate slots
    (OFFSET) put the masks in the sparse tail of the logit pattern -- a few blobs each, boxes of every size, many empty;
  * the hypernetworks' last layers are scaled (SCALE), which sets how many pixels lie within the stability offset of 1.0;
  * the IoU head's last bias moves per slot (IOU_SHIFT).
The constants were picked once from `--explore`; `main` asserts that the resulting counts stay in range, that no predicted
IoU lies within IOU_MARGIN of 0.88 and that every stability decision is at least STAB_SLACK_PX pixels from flipping -- so "same
records in the same order" is a fair demand of an fp32-accurate implementation with another summation order (threshold TIES
are covered by the tiny-model tests, where they are forced).  With random weights the three slots' masks are level sets of
three fixed patterns (the prompt moves the level, hardly the pattern), so the survivors are nested and the box NMS keeps a
ladder of box sizes per slot: ~500 candidates reach it, ~60 records leave it.

Stored: per record area / bbox / predicted_iou / stability_score / point_coords / candidate index, the id map (int32), the
packed masks of the first MASKS_STORED records, and of every candidate that passed the IoU filter its (iou, stability, box,
area) -- what a tolerance-mode comparison (fp16 SAM) needs to say which records moved and why.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from editanything_amd import arch, synth  # noqa: E402
from oracle import amg_oracle as AO, host_oracle  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
DECODER_SEED = 112
BIAS_CHANNEL = 0
ROW_SCALE = 0.1        # the prompt-dependent part of the offset (the hypernetworks' weight row of the bias channel)
# hypernetwork index = mask token index; multimask output uses tokens 1..3 (candidate slots 0..2)
OFFSET = {1: 0.0883, 2: -0.0545, 3: 0.0512}
SCALE = 40000.0
IOU_SHIFT = {1: 0.31255, 2: 1.4283, 3: 0.6833}
IOU_MARGIN = 2e-4      # |predicted IoU - 0.88| of every candidate (an fp32 implementation moves it by ~1e-6)
STAB_SLACK_PX = 0.04   # |#(logit > 1) - 0.95 #(logit > -1)| in PIXELS of every candidate that reaches the stability filter: the
                       # score is a ratio of two pixel counts, so its distance from 0.95 only means something in pixels; the
                       # counts are integers, so > 0 means "no exact tie: at least one pixel must flip to change the decision"
MASKS_STORED = 48


def calibrated_decoder_state_dict(seed=DECODER_SEED):
    """The synthetic prompt-encoder + mask-decoder state dict (upstream key names) with the calibration edit of the module
    docstring applied.  A pure function of the seed and the constants above: the GPU test regenerates exactly these tensors."""
    sd = {k: v.clone() for k, v in synth.synth_state_dict_torch(arch.sam_decoder_param_shapes(), seed).items()}
    c = BIAS_CHANNEL
    sd["mask_decoder.output_upscaling.3.weight"][:, c] = 0.0       # ConvTranspose2d weight [in, out, 2, 2]
    sd["mask_decoder.output_upscaling.3.bias"][c] = 1.0            # GELU(1) = 0.84134...
    for i in range(4):
        p = f"mask_decoder.output_hypernetworks_mlps.{i}.layers.2."
        sd[p + "weight"][c] *= ROW_SCALE
        if i in OFFSET:
            sd[p + "bias"][c] = float(OFFSET[i])
        sd[p + "weight"] *= float(SCALE)
        sd[p + "bias"] *= float(SCALE)
    for i, b in IOU_SHIFT.items():
        sd["mask_decoder.iou_prediction_head.layers.2.bias"][i] += float(b)
    return sd


def embedding():
    return torch.from_numpy(np.load(os.path.join(GOLD, "sam_vit_h_full.npz"))["embedding"]).float()


def explore():
    """Low-resolution (256^2) statistics of the 3072 candidates under the current constants."""
    emb = embedding()
    sd = calibrated_decoder_state_dict()
    pts = AO.build_point_grid(32) * np.array([[1024, 1024]])
    pe = AO.dense_pe(sd, emb.shape[-2:])
    lows, ious = [], []
    with torch.no_grad():
        for s in range(0, 1024, 64):
            p = torch.as_tensor(pts[s:s + 64], dtype=torch.float32)
            low, iou = AO.mask_decoder(sd, emb, pe, AO.embed_points(sd, p[:, None, :], torch.ones(len(p), 1), 1024), True)
            lows.append(low)
            ious.append(iou)
    low, iou = torch.cat(lows), torch.cat(ious)
    L = low.flatten(0, 1)
    inter = (L > 1).flatten(1).sum(1).float()
    union = (L > -1).flatten(1).sum(1).float()
    st = (inter / union).view(1024, 3)
    area = (L > 0).flatten(1).float().mean(1).view(1024, 3)
    boxes = AO.batched_mask_to_box(L > 0)
    for k in range(3):
        a, t, u = area[:, k], st[:, k], iou[:, k]
        m, sdev = low[:, k].mean((1, 2)), low[:, k].std((1, 2))
        q = torch.quantile(low[:, k].flatten(1)[:, ::7], 1 - 2e-4, dim=1)
        print(f"slot {k}: logit mean {float(m.mean()):+.4g} (sd over prompts {float(m.std()):.3g}), spatial sd {float(sdev.mean()):.3g}, "
              f"median 2e-4 quantile {float(q.median()):+.4g} (sd {float(q.std()):.3g}); empty {int((a == 0).sum())}, area q10/50/90 "
              f"{np.percentile(a.numpy(), [10, 50, 90])}, iou q10/50/90 {np.percentile(u.numpy(), [10, 50, 90])}, iou>0.88 {int((u > 0.88).sum())}, "
              f"stab>=0.95 {int((t >= 0.95).sum())}, both {int(((u > 0.88) & (t >= 0.95)).sum())}")
    k2 = ((iou > 0.88) & (st >= 0.95)).flatten()
    keep = AO.nms(boxes[k2].float(), iou.flatten()[k2], 0.7)
    bw = (boxes[k2][:, 2] - boxes[k2][:, 0]).float() / 256
    print(f"pass both {int(k2.sum())}, after NMS {len(keep)}; box width q10/50/90 of the passing {np.percentile(bw.numpy(), [10, 50, 90]) if len(bw) else None}")


def nms_with_margin(boxes, scores, thr):
    """AO.nms, also returning the smallest |IoU - thr| among the decisions the sweep actually takes (a kept box against every
    box still alive below it)."""
    order = torch.argsort(scores, descending=True, stable=True)
    iou = AO.box_iou(boxes[order], boxes[order])
    n = len(order)
    alive = torch.ones(n, dtype=torch.bool)
    keep, margin = [], 1.0
    for i in range(n):
        if not alive[i]:
            continue
        keep.append(int(order[i]))
        alive[i] = False
        if alive.any():
            margin = min(margin, float((iou[i][alive] - thr).abs().min()))
        alive &= ~(iou[i] > thr)
    return torch.as_tensor(keep, dtype=torch.long), margin


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    if "--explore" in sys.argv:
        return explore()
    emb = embedding()
    sd = calibrated_decoder_state_dict()
    H = W = 1024
    c = dict(AO.AMG_DEFAULTS)
    in_h, in_w = AO.preprocess_shape(H, W, 1024)
    pts = AO.build_point_grid(c["points_per_side"]) * np.array([[W, H]])
    pe = AO.dense_pe(sd, emb.shape[-2:])
    t0 = time.time()
    # the generator's loop (amg_oracle.generate), with the candidate index and the pre-filter numbers kept
    cand = dict(idx=[], iou=[], stab=[], box=[], area=[], slack=[])
    rec = dict(masks=[], iou=[], pts=[], stab=[], boxes=[], idx=[])
    n_cand = 0
    all_iou = []
    with torch.no_grad():
        for s in range(0, len(pts), c["points_per_batch"]):
            p = torch.as_tensor(pts[s:s + c["points_per_batch"]], dtype=torch.float32)
            tp = p * torch.tensor([in_w / W, in_h / H])
            low, iou = AO.mask_decoder(sd, emb, pe, AO.embed_points(sd, tp[:, None, :], torch.ones(len(p), 1), 1024), True)
            masks = AO.postprocess_masks(low, (in_h, in_w), (H, W), 1024).flatten(0, 1)
            iou = iou.flatten(0, 1)
            idx = torch.arange(3 * s, 3 * s + len(iou))
            n_cand += len(iou)
            all_iou.append(iou.clone())
            pp = p.repeat_interleave(3, dim=0)
            k = iou > c["pred_iou_thresh"]
            masks, iou, pp, idx = masks[k], iou[k], pp[k], idx[k]
            st = AO.stability_score(masks, c["mask_threshold"], c["stability_score_offset"])
            inter = (masks > c["mask_threshold"] + c["stability_score_offset"]).flatten(1).sum(1).double()
            union = (masks > c["mask_threshold"] - c["stability_score_offset"]).flatten(1).sum(1).double()
            cand["slack"].append((inter - c["stability_score_thresh"] * union).abs()[union > 0])
            mb = masks > c["mask_threshold"]
            boxes = AO.batched_mask_to_box(mb)
            cand["idx"].append(idx); cand["iou"].append(iou); cand["stab"].append(st); cand["box"].append(boxes)
            cand["area"].append(mb.flatten(1).sum(1))
            k = st >= c["stability_score_thresh"]
            k &= ~AO.is_box_near_crop_edge(boxes, [0, 0, W, H], [0, 0, W, H])
            rec["masks"].append(np.packbits(mb[k].numpy(), axis=-1)); rec["iou"].append(iou[k]); rec["pts"].append(pp[k])
            rec["stab"].append(st[k]); rec["boxes"].append(boxes[k]); rec["idx"].append(idx[k])
            print(f"prompts {s + len(p)}/1024: {int(k.sum())} of {len(iou)} IoU-passing candidates survive ({time.time() - t0:.0f} s)", flush=True)
    cand = {k: torch.cat(v) for k, v in cand.items()}
    packed = np.concatenate(rec["masks"])
    iou, ppts, stab, boxes, idx = (torch.cat(rec[k]) for k in ("iou", "pts", "stab", "boxes", "idx"))
    keep, nms_margin = nms_with_margin(boxes.float(), iou, c["box_nms_thresh"])
    assert torch.equal(keep, AO.nms(boxes.float(), iou, c["box_nms_thresh"]))
    n_iou, n_stab, n_rec = len(cand["iou"]), len(iou), len(keep)
    print(f"candidates {n_cand} -> IoU filter {n_iou} -> stability filter {n_stab} -> NMS {n_rec} records")
    # margins: every decision must be safe against a 1e-6-class perturbation of the logits
    m_iou = float((torch.cat(all_iou) - c["pred_iou_thresh"]).abs().min())
    m_stab = float(cand.pop("slack").min())
    print(f"margins: predicted IoU {m_iou:.2e}, stability {m_stab:.2f} px, NMS {nms_margin:.2e} (boxes are integers: they move only if an extreme pixel flips)")
    assert 60 <= n_rec <= 800, n_rec
    assert n_iou < 0.8 * n_cand and n_stab < 0.8 * n_iou and n_rec < 0.8 * n_stab, "every filter must do real work"
    assert m_iou >= IOU_MARGIN and m_stab >= STAB_SLACK_PX, (m_iou, m_stab)
    records = []
    for i in keep.tolist():
        seg = np.unpackbits(packed[i], axis=-1)[:, :W].astype(bool)
        b = boxes[i].tolist()
        records.append(dict(segmentation=seg, area=int(seg.sum()), bbox=[b[0], b[1], b[2] - b[0], b[3] - b[1]],
                            predicted_iou=float(iou[i]), point_coords=[ppts[i].tolist()], stability_score=float(stab[i]),
                            crop_box=[0, 0, W, H]))
    idmap = host_oracle.show_anns_idmap(records)
    out = dict(
        n_candidates=n_cand, n_pass_iou=n_iou, n_pass_stability=n_stab, n_records=n_rec,
        rec_candidate=idx[keep].numpy().astype(np.int32), rec_area=np.array([r["area"] for r in records], np.int64),
        rec_bbox=np.array([r["bbox"] for r in records], np.int32), rec_iou=iou[keep].numpy(), rec_stability=stab[keep].numpy(),
        rec_point=ppts[keep].numpy(), rec_masks_packed=packed[keep[:MASKS_STORED].numpy()],
        cand_index=cand["idx"].numpy().astype(np.int32), cand_iou=cand["iou"].numpy(), cand_stability=cand["stab"].numpy(),
        cand_box=cand["box"].numpy().astype(np.int32), cand_area=cand["area"].numpy().astype(np.int64),
        idmap=np.asarray(idmap), margin_iou=m_iou, margin_stability_px=m_stab, margin_nms=nms_margin,
        calibration=np.array([DECODER_SEED, BIAS_CHANNEL, ROW_SCALE, SCALE] + [OFFSET[i] for i in (1, 2, 3)] + [IOU_SHIFT[i] for i in (1, 2, 3)], np.float64))
    path = os.path.join(GOLD, "amg_vith_full.npz")
    np.savez_compressed(path, **out)
    print("written", path, f"{os.path.getsize(path) / 1e6:.1f} MB; id map: {len(np.unique(np.asarray(idmap).reshape(-1, np.asarray(idmap).shape[-1]), axis=0))} distinct values")


if __name__ == "__main__":
    main()
