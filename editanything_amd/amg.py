"""SAM prompt encoder + mask decoder + automatic mask generation on the MI355X.

Call surface of the third-party objects the reference uses:
  * `SamAutomaticMaskGenerator(sam).generate(image) -> list[dict]`   (sam2image.py:71,118; editany_lora.py:523)
  * `SamPredictor(sam).set_image(image)` / `.predict(point_coords, point_labels, multimask_output)`
    (editany_lora.py:527-543)
with upstream's defaults (points_per_side 32, points_per_batch 64, pred_iou_thresh 0.88, stability_score_thresh 0.95,
stability_score_offset 1.0, box_nms_thresh 0.7, crop_n_layers 0, min_mask_region_area 0, "binary_mask" output).
Record order = NMS keep order (descending predicted_iou), which is the order `show_anns` indexes.

Device work: the image encoder is `editanything_amd.sam.ImageEncoderViT`; in the decoder every contraction over the
4096 image tokens of a point batch (k/v/q projections of the two-way transformer's image side, out-projection +
residual, both transposed-conv upscalers as GEMMs with a fused GELU) runs through the C-ABI MFMA kernels (fp16
operands, fp32 accumulate); LayerNorms through `ea_layernorm_f16`.  The 7-token side (self-attention, token MLPs,
hypernetwork heads) are short torch-on-device expressions; the mask post-processing (both bilinear resizes of
`postprocess_masks`, threshold, stability counts, boxes) is ONE C-ABI pass (`ea_sam_mask_postprocess`) that writes
only the 1-byte masks instead of materialising the 1024^2 fp32 upsampling of every candidate; box NMS computes the IoU
matrix on the device and sweeps the few hundred candidates on the host.
The encoder and decoder run fp16 where the reference runs fp32: masks agree with the fp32 oracle to a few pixels per
mask on the `logits > 0` boundary (tests/test_amg.py states the tolerance); `show_anns` itself stays bit-exact.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import ops

DECODE_BATCH = 1024  # prompts per decoder call (keys [1024, 4096, 256] fp16 = 2 GB, ~13 GB of activations in all: the
                     # whole 32 x 32 grid in one call; results do not depend on it)

AMG_DEFAULTS = dict(points_per_side=32, points_per_batch=64, pred_iou_thresh=0.88, stability_score_thresh=0.95,
                    stability_score_offset=1.0, box_nms_thresh=0.7, mask_threshold=0.0)


def _f16(t, dev):
    return t.to(device=dev, dtype=torch.float16).contiguous()


def _f32(t, dev):
    return t.to(device=dev, dtype=torch.float32).contiguous()


class _Attn:
    """segment_anything modeling/transformer.py Attention (q/k/v/out projections, `heads` heads over the internal dim)."""

    def __init__(self, sd, p, dev, heads):
        self.heads = heads
        self.w = {n: _f16(sd[f"{p}{n}_proj.weight"], dev) for n in "qkv"}
        self.b = {n: _f32(sd[f"{p}{n}_proj.bias"], dev) for n in "qkv"}
        self.wo, self.bo = _f16(sd[p + "out_proj.weight"], dev), _f32(sd[p + "out_proj.bias"], dev)
        self.internal = self.wo.shape[1]

    def proj(self, n, x16):
        return ops.gemm(x16, self.w[n], self.b[n])

    def core(self, q, k, v):
        """softmax(q k^T / sqrt(d)) v on already projected [B, N, internal] fp16 tensors (fp32 scores)."""
        B, Nq, Ci = q.shape
        h, d = self.heads, Ci // self.heads
        sp = lambda t: t.reshape(B, t.shape[1], h, d).transpose(1, 2)
        if Nq * k.shape[1] >= 4096:
            # image side (4096 queries x 7 keys or 7 x 4096): fused fp16 attention, fp32 softmax inside; no fp32 copies of
            # the [B, 4096, C] operands
            o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v), scale=1.0 / math.sqrt(d))
            return o.transpose(1, 2).reshape(B, Nq, Ci)
        if h == 8 and Ci == 256 and Nq <= 8 and k.shape[1] == Nq and q.is_cuda:
            # the prompt tokens' self attention: one launch, fp32 inside (round 6)
            return ops.sam_token_self_attn(q.contiguous(), k.contiguous(), v.contiguous(), 1.0 / math.sqrt(d))
        a = torch.matmul(sp(q).float(), sp(k).float().transpose(-2, -1)) * (1.0 / math.sqrt(d))
        o = torch.matmul(torch.softmax(a, dim=-1), sp(v).float())
        return o.transpose(1, 2).reshape(B, Nq, Ci).half()


class SamPromptDecoder:
    """PromptEncoder (points, no mask prompt) + MaskDecoder.  `state_dict`: a SAM checkpoint's `prompt_encoder.*` and
    `mask_decoder.*` entries (editanything_amd.arch.sam_decoder_param_shapes)."""

    def __init__(self, state_dict, device="cuda", heads=8, img_size=1024):
        sd, dev = state_dict, torch.device(device)
        self.device, self.img_size, self.heads = dev, img_size, heads
        pe = "prompt_encoder."
        self.gauss = _f32(sd[pe + "pe_layer.positional_encoding_gaussian_matrix"], dev)
        self.point_emb = [_f32(sd[f"{pe}point_embeddings.{i}.weight"], dev) for i in range(4)]
        self.not_a_point = _f32(sd[pe + "not_a_point_embed.weight"], dev)
        self.no_mask = _f32(sd[pe + "no_mask_embed.weight"], dev)
        md = "mask_decoder."
        self.out_tokens = torch.cat([_f32(sd[md + "iou_token.weight"], dev), _f32(sd[md + "mask_tokens.weight"], dev)], 0)
        self.C = self.out_tokens.shape[1]
        t = md + "transformer."
        self.layers = []
        i = 0
        while f"{t}layers.{i}.self_attn.q_proj.weight" in sd:
            lp = f"{t}layers.{i}."
            self.layers.append(dict(
                self_attn=_Attn(sd, lp + "self_attn.", dev, heads), t2i=_Attn(sd, lp + "cross_attn_token_to_image.", dev, heads),
                i2t=_Attn(sd, lp + "cross_attn_image_to_token.", dev, heads),
                norms=[(_f32(sd[f"{lp}norm{k}.weight"], dev), _f32(sd[f"{lp}norm{k}.bias"], dev)) for k in (1, 2, 3, 4)],
                w1=_f16(sd[lp + "mlp.lin1.weight"], dev), b1=_f32(sd[lp + "mlp.lin1.bias"], dev),
                w2=_f16(sd[lp + "mlp.lin2.weight"], dev), b2=_f32(sd[lp + "mlp.lin2.bias"], dev)))
            i += 1
        self.final = _Attn(sd, t + "final_attn_token_to_image.", dev, heads)
        self.norm_final = (_f32(sd[t + "norm_final_attn.weight"], dev), _f32(sd[t + "norm_final_attn.bias"], dev))
        # ConvTranspose2d(k=2, s=2) == one GEMM per input pixel: N = (dy, dx, cout), K = cin
        w0 = sd[md + "output_upscaling.0.weight"]                       # [cin, cout, 2, 2]
        self.up0_w = _f16(w0.permute(2, 3, 1, 0).reshape(-1, w0.shape[0]), dev)
        self.up0_b = _f32(sd[md + "output_upscaling.0.bias"].repeat(4), dev)
        self.up_ln = (_f32(sd[md + "output_upscaling.1.weight"], dev), _f32(sd[md + "output_upscaling.1.bias"], dev))
        w1 = sd[md + "output_upscaling.3.weight"]
        self.up1_w = _f16(w1.permute(2, 3, 1, 0).reshape(-1, w1.shape[0]), dev)
        self.up1_b = _f32(sd[md + "output_upscaling.3.bias"].repeat(4), dev)
        self.c0, self.c1 = w0.shape[1], w1.shape[1]
        mlp = lambda p: [(_f32(sd[f"{p}layers.{j}.weight"], dev), _f32(sd[f"{p}layers.{j}.bias"], dev)) for j in range(3)]
        self.hyper = [mlp(f"{md}output_hypernetworks_mlps.{i}.") for i in range(self.out_tokens.shape[0] - 1)]
        self.iou_head = mlp(md + "iou_prediction_head.")
        self._pe_cache = {}
        self._graphs = {}          # predict_masks_graph: (shapes, scratch tag) -> (graph, static inputs, outputs, scratch refs)
        self.graph_ok = True       # cleared for good if a capture ever fails (the eager path is the same arithmetic)

    # ------------------------------------------------------------------ prompt encoder
    def _pe(self, coords01):
        c = (2.0 * coords01 - 1.0) @ self.gauss
        c = 2.0 * math.pi * c
        return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)

    def dense_pe(self, size):
        """PromptEncoder.get_dense_pe as image tokens: fp16 [h*w, C]."""
        if size not in self._pe_cache:
            h, w = size
            y = (torch.arange(h, dtype=torch.float32, device=self.device) + 0.5) / h
            x = (torch.arange(w, dtype=torch.float32, device=self.device) + 0.5) / w
            grid = torch.stack([x[None, :].expand(h, w), y[:, None].expand(h, w)], dim=-1)
            self._pe_cache[size] = self._pe(grid).reshape(h * w, -1).half().contiguous()
        return self._pe_cache[size]

    def embed_points(self, points, labels):
        """points [B, N, 2] (x, y) in the 1024-frame, labels [B, N] -> sparse embeddings fp32 [B, N + 1, C]."""
        B = points.shape[0]
        pts = torch.cat([points.to(self.device).float() + 0.5, torch.zeros(B, 1, 2, device=self.device)], dim=1)
        lab = torch.cat([labels.to(self.device).float(), -torch.ones(B, 1, device=self.device)], dim=1)[..., None]
        emb = self._pe(pts / float(self.img_size))
        emb = torch.where(lab == -1, torch.zeros_like(emb), emb)
        return emb + (lab == -1) * self.not_a_point + (lab == 0) * self.point_emb[0] + (lab == 1) * self.point_emb[1]

    def embed_boxes(self, boxes):
        """PromptEncoder._embed_boxes: boxes [B, 4] XYXY in the input frame -> sparse embeddings fp32 [B, 2, C] (corner
        encodings + point_embeddings[2] / [3]; no padding point when a box is given)."""
        c = (boxes.to(self.device).float() + 0.5).reshape(-1, 2, 2)
        emb = self._pe(c / float(self.img_size))
        return emb + torch.cat([self.point_emb[2], self.point_emb[3]], dim=0)[None]

    # ------------------------------------------------------------------ mask decoder
    @staticmethod
    def _mlp3(layers, x):
        for j, (w, b) in enumerate(layers):
            x = F.linear(x, w, b)
            if j < 2:
                x = F.relu(x)
        return x

    def _ln(self, x, wb, eps=1e-5):
        return F.layer_norm(x, (x.shape[-1],), wb[0], wb[1], eps)

    def predict_masks_unfused(self, image_tokens, emb_hw, sparse, multimask_output=True):
        """(The round-1/2 formulation: kept as the A/B reference of `predict_masks`, tests/test_amg.py.)
        image_tokens: fp16 [h*w, C] (encoder output + no-mask dense embedding, NHWC order); sparse fp32 [B, Np, C].
        -> low-res mask logits fp32 [B, 3|1, 4h, 4w], iou predictions fp32 [B, 3|1]."""
        B, C = sparse.shape[0], self.C
        h, w = emb_hw
        T = h * w
        key_pe = self.dense_pe(emb_hw)                                   # [T, C] fp16
        point_emb = torch.cat([self.out_tokens[None].expand(B, -1, -1), sparse], dim=1)   # fp32 [B, 7, C]
        queries = point_emb
        keys = image_tokens[None].expand(B, -1, -1)                     # fp16 [B, T, C] (layer 0: shared by all points)
        shared = True
        for li, L in enumerate(self.layers):
            a = L["self_attn"]
            if li == 0:
                q16 = queries.half()
                queries = F.linear(a.core(a.proj("q", q16), a.proj("k", q16), a.proj("v", q16)).float(), a.wo.float(), a.bo)
            else:
                q16 = (queries + point_emb).half()
                att = a.core(a.proj("q", q16), a.proj("k", q16), a.proj("v", queries.half()))
                queries = queries + F.linear(att.float(), a.wo.float(), a.bo)
            queries = self._ln(queries, L["norms"][0])
            # tokens -> image.  In the first layer the image side is the same for every point: project it once.
            a = L["t2i"]
            src = keys[0] if shared else keys
            kin = (src + key_pe).contiguous() if shared else (src + key_pe[None]).reshape(B * T, C)
            k = a.proj("k", kin).reshape(1 if shared else B, T, -1)
            v = a.proj("v", src.reshape(-1, C)).reshape(1 if shared else B, T, -1)
            if shared:
                k, v = k.expand(B, -1, -1), v.expand(B, -1, -1)
            att = a.core(a.proj("q", (queries + point_emb).half()), k, v)
            queries = self._ln(queries + F.linear(att.float(), a.wo.float(), a.bo), L["norms"][1])
            m = F.linear(F.relu(F.linear(queries, L["w1"].float(), L["b1"])), L["w2"].float(), L["b2"])
            queries = self._ln(queries + m, L["norms"][2])
            # image -> tokens: the image side becomes per-point from here on
            a = L["i2t"]
            qt = (queries + point_emb).half()
            qi = a.proj("q", kin).reshape(1 if shared else B, T, -1)
            if shared:
                qi = qi.expand(B, -1, -1)
            att = a.core(qi, a.proj("k", qt), a.proj("v", queries.half()))       # [B, T, internal]
            res = keys.reshape(B * T, C) if not shared else keys.contiguous().reshape(B * T, C)
            keys = ops.gemm(att.reshape(B * T, -1), a.wo, a.bo, residual=res)
            keys = ops.layernorm(keys, L["norms"][3][0], L["norms"][3][1], eps=1e-5).reshape(B, T, C)
            shared = False
        a = self.final
        kin = (keys + key_pe[None]).reshape(B * T, C)
        att = a.core(a.proj("q", (queries + point_emb).half()), a.proj("k", kin).reshape(B, T, -1),
                     a.proj("v", keys.reshape(B * T, C)).reshape(B, T, -1))
        queries = self._ln(queries + F.linear(att.float(), a.wo.float(), a.bo), self.norm_final)
        iou_tok, mask_toks = queries[:, 0], queries[:, 1:1 + len(self.hyper)]
        # upscaling: two stride-2 transposed convs == per-pixel GEMMs + pixel shuffle (NHWC)
        u = ops.gemm(keys.reshape(B * T, C), self.up0_w, self.up0_b)                                  # [B*T, 4*c0]
        u = u.reshape(B, h, w, 2, 2, self.c0).permute(0, 1, 3, 2, 4, 5).reshape(B * 4 * T, self.c0)
        u = F.gelu(ops.layernorm(u, self.up_ln[0], self.up_ln[1], eps=1e-6))      # fp16 in / out, evaluated in fp32
        u = ops.gemm(u, self.up1_w, self.up1_b, act=ops.ACT_GELU)                                     # [B*4T, 4*c1]
        u = u.reshape(B, 2 * h, 2 * w, 2, 2, self.c1).permute(0, 1, 3, 2, 4, 5).reshape(B, 16 * T, self.c1)
        hyper = torch.stack([self._mlp3(self.hyper[i], mask_toks[:, i]) for i in range(len(self.hyper))], dim=1)
        # [B, 16T, c1] x [B, c1, 4] on the fp16 feature map (fp32 accumulate in the library GEMM; no fp32 copy of u)
        masks = torch.bmm(u, hyper.half().transpose(1, 2)).float().transpose(1, 2).reshape(B, -1, 4 * h, 4 * w)
        iou = self._mlp3(self.iou_head, iou_tok)
        sl = slice(1, None) if multimask_output else slice(0, 1)
        return masks[:, sl], iou[:, sl]

    # ---- the image-token side restated per token (csrc/ea_sam.hip): what crosses the ABI per block
    def _heads(self, a):
        """An _Attn module's projections as per-head blocks: Wq/Wk/Wv [h, d, C], biases [h, d], Wo [C, h, d]."""
        if not hasattr(a, "_blocks"):
            h, Ci = self.heads, a.internal
            d = Ci // h
            a._blocks = dict(d=d, wq=a.w["q"].float().reshape(h, d, -1).contiguous(), wk=a.w["k"].float().reshape(h, d, -1).contiguous(),
                             wv=a.w["v"].float().reshape(h, d, -1), bq=a.b["q"].reshape(h, d), bk=a.b["k"].reshape(h, d),
                             bv=a.b["v"].reshape(h, d), wo=a.wo.float().reshape(-1, h, d),
                             # the layouts ops.sam_fold_heads / sam_unfold_heads read: Wo as [h, d, C], Wv transposed [C, h * d]
                             wo_hdc=a.wo.float().reshape(-1, h, d).permute(1, 2, 0).contiguous(),
                             wv_t=a.w["v"].float().t().contiguous(), bv_flat=a.b["v"].float().contiguous())
        return a._blocks

    def _t2i_folded(self, a, q_in, kp, k, key_pe):
        """Token -> image cross attention with PER-PROMPT keys, without projecting the image side: scores = G (keys + pe)^T
        with G[h, j] = Wk_h^T q_hj (the key bias is constant over the image tokens and drops out of their softmax), output
        = Wv_h (P keys) + bv_h.  kp / k fp16 [B, T, C]; q_in fp32 [B, 7, C] -> fp32 [B, 7, C] (before the residual)."""
        bl = self._heads(a)
        B, n, Cc = q_in.shape
        h, d, T = self.heads, bl["d"], k.shape[-2]
        qh = F.linear(q_in, a.w["q"].float(), a.b["q"])                   # [B, n, h * d]
        fused = d in (16, 32) and h == 8 and Cc == 256
        if fused:
            G = ops.sam_fold_heads(qh.contiguous(), bl["wk"]).reshape(B, h, 8, Cc)        # one launch (round 6), rows j >= n zero
        else:
            G = torch.zeros((B, h, 8, Cc), dtype=torch.float16, device=q_in.device)
            G[:, :, :n] = torch.einsum("bjhd,hdc->bhjc", qh.reshape(B, n, h, d), bl["wk"]).half()
        if T % 64 == 0:
            # one pass over the image tokens per prompt: scores, online softmax and P keys fused (ops.sam_t2i)
            ctx = ops.sam_t2i(k, key_pe, G.reshape(B, 64, Cc), 1.0 / math.sqrt(d), B)
            if fused:
                # ... and back through the value projection in one launch: o = Wv_h ctx_hj + bv_h
                return F.linear(ops.sam_unfold_heads(ctx, bl["wv_t"], bl["bv_flat"], n), a.wo.float(), a.bo)
            ctx = ctx.reshape(B, h, 8, Cc)[:, :, :n]
        elif k.dim() == 2:
            # block 0: ONE image-token tensor for every prompt -- two plain contractions over all B * 64 score rows
            S = ops.gemm(G.reshape(B * 64, Cc), kp, out_dtype=torch.float32).reshape(B, 64, T)
            P = ops.softmax_rows(S, 1.0 / math.sqrt(d))
            ctx = ops.gemm(P.reshape(B * 64, T), k.t().contiguous()).float().reshape(B, h, 8, Cc)[:, :, :n]
        else:
            # per-prompt keys: G (keys + pe)^T = G keys^T + G pe^T -- the second product has ONE weight for all prompts (a plain
            # contraction) and enters the batched one as its fp32 residual, so no keys + pe tensor is ever stored
            S_pe = ops.gemm(G.reshape(B * 64, Cc), key_pe, out_dtype=torch.float32)
            S = torch.empty((B, 64, T), dtype=torch.float32, device=q_in.device)
            ops.gemm_batched(G, k, S, 64, T, Cc, B, 64 * Cc, T * Cc, 64 * T, residual=S_pe, stride_r=64 * T)
            P = ops.softmax_rows(S, 1.0 / math.sqrt(d))                               # fp16 [B, 64, T]
            ctx = torch.bmm(P, k).float().reshape(B, h, 8, Cc)[:, :, :n]              # [B, h, 7, C]
        o = torch.einsum("bhjc,hdc->bjhd", ctx, bl["wv"]) + bl["bv"]
        return F.linear(o.reshape(B, n, h * d), a.wo.float(), a.bo)

    def _i2t_fused(self, a, q_pe, q_val, kp, k, key_pe, norm, B, want_kp):
        """Image -> token cross attention + residual + norm4, one pass (ops.sam_i2t).  q_pe = queries + point embedding
        (the attention's keys), q_val = queries (its values), fp32 [B, 7, C]; kp / k fp16 [T, C] or [B, T, C]."""
        bl = self._heads(a)
        n, Cc = q_pe.shape[1], q_pe.shape[2]
        h, d = self.heads, bl["d"]
        scale = 1.0 / math.sqrt(d)
        kh = F.linear(q_pe, a.w["k"].float(), a.b["k"]).reshape(B, n, h, d)
        vh = F.linear(q_val, a.w["v"].float(), a.b["v"]).reshape(B, n, h, d)
        cb = torch.full((B, h, 8), -1e30, dtype=torch.float32, device=q_pe.device)
        cb[:, :, :n] = torch.einsum("bjhd,hd->bhj", kh, bl["bq"]) * scale
        if d in (16, 32) and h == 8 and Cc == 256:
            # round 6: each operand in one launch (ops.sam_fold_heads) instead of einsum + cast + zero fill + strided copy
            g2 = ops.sam_fold_heads(kh.reshape(B, n, h * d), bl["wq"])
            vo = ops.sam_fold_heads(vh.reshape(B, n, h * d), bl["wo_hdc"], perm=self._vo_perm_i32(), c_major=True)
            return ops.sam_i2t(kp, k, key_pe, g2, cb.reshape(B, 64), vo, a.bo, norm[0], norm[1], 1e-5, scale, B, want_kp)
        g2 = torch.zeros((B, h, 8, Cc), dtype=torch.float16, device=q_pe.device)
        g2[:, :, :n] = torch.einsum("bjhd,hdc->bhjc", kh, bl["wq"]).half()
        # vo[b, c, s] = Wo_h v_hj for the score column (h, j) = perm[s] that storage position s holds: gather the SMALL
        # operands into that order ([B, 64, d] values, [C, 64, d] weights), one contraction writes the layout the kernel reads
        vpad = torch.zeros((B, h, 8, d), dtype=torch.float32, device=q_pe.device)
        vpad[:, :, :n] = vh.permute(0, 2, 1, 3)
        perm = self._vo_perm()
        v_s = vpad.reshape(B, 64, d)[:, perm]                              # [B, 64, d]
        wo_s = bl["wo"][:, perm // 8]                                      # [C, 64, d]
        vo = torch.einsum("bsd,csd->bcs", v_s, wo_s).half().contiguous()
        return ops.sam_i2t(kp, k, key_pe, g2.reshape(B, 64, Cc), cb.reshape(B, 64), vo, a.bo, norm[0], norm[1], 1e-5, scale, B, want_kp)

    def _vo_perm(self):
        if not hasattr(self, "_perm"):
            self._perm = ops.sam_vo_perm(self.device)
            self._perm_i32 = self._perm.int()
        return self._perm

    def _vo_perm_i32(self):
        self._vo_perm()
        return self._perm_i32

    @torch.no_grad()
    def predict_masks_graph(self, image_tokens, emb_hw, sparse, multimask_output=True):
        """`predict_masks` replayed from a HIP graph captured once per (shapes, scratch tag): the automatic generator decodes
        the SAME 1024 grid prompts for every image -- ~270 launches, two thirds of them small torch ops of the 7-token side,
        become one replay (like sam.ImageEncoderViT.forward_graph: thread-local capture, the entry owns the scratch it
        addresses).  The returned tensors are the graph's own outputs: valid until the next call with the same shapes."""
        if not self.graph_ok or not image_tokens.is_cuda:
            return self.predict_masks(image_tokens, emb_hw, sparse, multimask_output)
        key = (tuple(image_tokens.shape), tuple(emb_hw), tuple(sparse.shape), bool(multimask_output), ops.aux_tag())
        ent = self._graphs.get(key)
        if ent is None:
            tok_s, sp_s = image_tokens.clone(), sparse.clone()
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                self.predict_masks(tok_s, emb_hw, sp_s, multimask_output)      # warm-up outside capture (lazy caches, library handles)
            cur.wait_stream(side)
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    out = self.predict_masks(tok_s, emb_hw, sp_s, multimask_output)
            except Exception as exc:                                           # noqa: BLE001 -- any capture failure: eager from now on
                import warnings
                warnings.warn(f"SamPromptDecoder: graph capture of the mask decoder failed ({exc}); running it eagerly")
                self.graph_ok = False
                torch.cuda.synchronize()
                return self.predict_masks(image_tokens, emb_hw, sparse, multimask_output)
            ent = self._graphs[key] = (g, tok_s, sp_s, out, ops.workspace_refs())
        g, tok_s, sp_s, out = ent[:4]
        tok_s.copy_(image_tokens)
        sp_s.copy_(sparse)
        g.replay()
        return out

    def predict_masks(self, image_tokens, emb_hw, sparse, multimask_output=True):
        """image_tokens: fp16 [h*w, C] (encoder output + no-mask dense embedding, NHWC order); sparse fp32 [B, Np, C].
        -> low-res mask logits fp32 [B, 3|1, 4h, 4w], iou predictions fp32 [B, 3|1].

        segment_anything's MaskDecoder.predict_masks / TwoWayTransformer (third party).  The image side ([B, 4096, 256]
        per block) is touched in as few passes as the arithmetic allows: the image -> token attention + residual + norm4
        of a block is one fused kernel, the token -> image attention folds the key / value projections into the 7-token
        side (no projection of the image tokens at all), and everything after the first transposed conv is one kernel."""
        if sparse.shape[1] + self.out_tokens.shape[0] > 7:
            return self.predict_masks_unfused(image_tokens, emb_hw, sparse, multimask_output)   # > 2 prompt tokens: 8 score columns per head
        B, C = sparse.shape[0], self.C
        h, w = emb_hw
        T = h * w
        key_pe = self.dense_pe(emb_hw)                                   # [T, C] fp16
        point_emb = torch.cat([self.out_tokens[None].expand(B, -1, -1), sparse], dim=1)   # fp32 [B, n <= 7, C]
        queries = point_emb
        k, kp = image_tokens, (image_tokens.float() + key_pe.float()).half()               # shared by every prompt in block 0
        for li, L in enumerate(self.layers):
            a = L["self_attn"]
            if li == 0:
                q16 = queries.half()
                queries = F.linear(a.core(a.proj("q", q16), a.proj("k", q16), a.proj("v", q16)).float(), a.wo.float(), a.bo)
            else:
                q16 = (queries + point_emb).half()
                att = a.core(a.proj("q", q16), a.proj("k", q16), a.proj("v", queries.half()))
                queries = queries + F.linear(att.float(), a.wo.float(), a.bo)
            queries = self._ln(queries, L["norms"][0])
            att = self._t2i_folded(L["t2i"], queries + point_emb, kp, k, key_pe)
            queries = self._ln(queries + att, L["norms"][1])
            m = F.linear(F.relu(F.linear(queries, L["w1"].float(), L["b1"])), L["w2"].float(), L["b2"])
            queries = self._ln(queries + m, L["norms"][2])
            k, _ = self._i2t_fused(L["i2t"], queries + point_emb, queries, kp if k.dim() == 2 else None, k, key_pe, L["norms"][3], B,
                                   False)
        att = self._t2i_folded(self.final, queries + point_emb, None, k, key_pe)
        queries = self._ln(queries + att, self.norm_final)
        iou_tok, mask_toks = queries[:, 0], queries[:, 1:1 + len(self.hyper)]
        hyper = torch.stack([self._mlp3(self.hyper[i], mask_toks[:, i]) for i in range(len(self.hyper))], dim=1).contiguous()
        nh = len(self.hyper)
        m0, nm = (1, nh - 1) if multimask_output else (0, 1)             # MaskDecoder.forward's slice, written densely
        masks = torch.empty((B, nm, 4 * h, 4 * w), dtype=torch.float32, device=k.device)
        if T % 16 == 0 and C == 256 and self.c0 == 64 and self.c1 == 32:
            # round 6: output_upscaling whole in ONE pass over the image tokens (the first transposed conv's [B*T*4, 64] output --
            # 2 GB for the 1024 grid prompts -- is never stored)
            ops.sam_upscale(k.reshape(B * T, C), self.up0_w, self.up0_b, self.up_ln[0], self.up_ln[1], 1e-6, self.up1_w, self.up1_b,
                            hyper, B, h, w, m0, nm, out=masks)
            iou = self._mlp3(self.iou_head, iou_tok)
            return masks, iou[:, m0:m0 + nm]
        step = max(1, (1 << 30) // (T * C * 2))                          # operands of one launch stay under the 2-GiB buffer range
        for b0 in range(0, B, step):
            b1 = min(B, b0 + step)
            u0 = ops.gemm(k[b0:b1].reshape((b1 - b0) * T, C), self.up0_w, self.up0_b)              # [b*T, 4*c0]
            ops.sam_upscale_tail(u0.reshape((b1 - b0) * T * 4, self.c0), self.up_ln[0], self.up_ln[1], 1e-6, self.up1_w, self.up1_b,
                                 hyper[b0:b1].contiguous(), b1 - b0, h, w, m0, nm, out=masks[b0:b1])
        iou = self._mlp3(self.iou_head, iou_tok)
        return masks, iou[:, m0:m0 + nm]

    def image_tokens(self, embedding_nchw):
        """Encoder output [1, C, h, w] -> decoder image tokens (embedding + no-mask dense prompt) fp16 [h*w, C]."""
        e = embedding_nchw.to(self.device).float()
        return (e[0].permute(1, 2, 0).reshape(-1, e.shape[1]) + self.no_mask).half().contiguous()


# ---------------------------------------------------------------------- post-processing (utils/amg.py, on device)
def preprocess_shape(oldh, oldw, long_side=1024):
    scale = long_side * 1.0 / max(oldh, oldw)
    return int(oldh * scale + 0.5), int(oldw * scale + 0.5)


def build_point_grid(n):
    off = 1.0 / (2 * n)
    pts = np.linspace(off, 1 - off, n)
    return np.stack([np.tile(pts[None, :], (n, 1)), np.tile(pts[:, None], (1, n))], axis=-1).reshape(-1, 2)


def postprocess_masks(masks, input_size, original_size, img_size=1024):
    masks = F.interpolate(masks, (img_size, img_size), mode="bilinear", align_corners=False)
    masks = masks[..., :input_size[0], :input_size[1]]
    return F.interpolate(masks, original_size, mode="bilinear", align_corners=False)


def stability_score(masks, thr, off):
    inter = (masks > (thr + off)).sum(-1, dtype=torch.int16).sum(-1, dtype=torch.int32)
    union = (masks > (thr - off)).sum(-1, dtype=torch.int16).sum(-1, dtype=torch.int32)
    return inter / union


def batched_mask_to_box(masks):
    if masks.numel() == 0:
        return torch.zeros(*masks.shape[:-2], 4, device=masks.device)
    h, w = masks.shape[-2:]
    dev = masks.device
    in_h, _ = torch.max(masks, dim=-1)
    hc = in_h * torch.arange(h, device=dev)[None, :]
    bottom, _ = torch.max(hc, dim=-1)
    top, _ = torch.min(hc + h * (~in_h), dim=-1)
    in_w, _ = torch.max(masks, dim=-2)
    wc = in_w * torch.arange(w, device=dev)[None, :]
    right, _ = torch.max(wc, dim=-1)
    left, _ = torch.min(wc + w * (~in_w), dim=-1)
    empty = (right < left) | (bottom < top)
    return torch.stack([left, top, right, bottom], dim=-1) * (~empty).unsqueeze(-1)


def is_box_near_crop_edge(boxes, crop_box, orig_box, atol=20.0):
    cb = torch.as_tensor(crop_box, dtype=torch.float, device=boxes.device)
    ob = torch.as_tensor(orig_box, dtype=torch.float, device=boxes.device)
    b = boxes.float()
    near_crop = torch.isclose(b, cb[None, :], atol=atol, rtol=0)
    near_img = torch.isclose(b, ob[None, :], atol=atol, rtol=0)
    return torch.any(near_crop & ~near_img, dim=1)


def nms(boxes, scores, thr):
    """torchvision.ops.nms semantics; the IoU matrix is computed on the device, the greedy sweep over the (few hundred)
    candidates on the host.  Returns kept indices in descending-score order."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.long)
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order].float()
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(b[:, None, :2], b[None, :, :2])
    rb = torch.min(b[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    over = ((inter / (area[:, None] + area[None, :] - inter)) > thr).cpu().numpy()
    order = order.cpu().numpy()
    alive = np.ones(n, dtype=bool)
    keep = []
    for i in range(n):
        if not alive[i]:
            continue
        keep.append(int(order[i]))
        alive &= ~over[i]
        alive[i] = False
    return torch.as_tensor(keep, dtype=torch.long)


class SamAutomaticMaskGenerator:
    """`SamAutomaticMaskGenerator(sam)` of the reference (sam2image.py:71): `encoder` is an
    `editanything_amd.sam.ImageEncoderViT`, `decoder` a `SamPromptDecoder`; keyword arguments override upstream's
    defaults under their upstream names."""

    def __init__(self, encoder, decoder, decode_batch=None, use_graph=True, **overrides):
        """decode_batch: prompts per decoder call (default DECODE_BATCH = 1024: the whole 32 x 32 grid at once, ~13 GB of
        activations).  It is this implementation's memory knob; upstream's `points_per_batch` is honoured as a LOWER bound of
        it only (prompts are independent: neither changes a result) -- pass decode_batch=64 on a small device."""
        self.encoder, self.decoder = encoder, decoder
        self.decode_batch = DECODE_BATCH if decode_batch is None else max(1, int(decode_batch))
        self.use_graph = bool(use_graph)     # the grid's decoder pass as one HIP-graph replay (SamPromptDecoder.predict_masks_graph)
        self._sparse_cache = {}
        self.cfg = dict(AMG_DEFAULTS)
        unknown = set(overrides) - set(self.cfg)
        if unknown:
            raise TypeError(f"unsupported generator arguments: {sorted(unknown)}")
        self.cfg.update(overrides)

    def set_image(self, image_u8_hwc):
        """SamPredictor.set_image: ResizeLongestSide(1024) -> normalise / pad -> encoder.  Returns decoder tokens."""
        from . import host
        img = np.asarray(image_u8_hwc)
        H, W = img.shape[:2]
        S = self.encoder.cfg["img_size"]
        in_h, in_w = preprocess_shape(H, W, S)
        resized = host.resize_longest_side(img, S) if (H, W) != (in_h, in_w) else img
        emb = self.encoder.encode_image(resized)
        self._state = dict(tokens=self.decoder.image_tokens(emb), emb_hw=tuple(emb.shape[-2:]), orig=(H, W), inp=(in_h, in_w))
        return self._state

    @torch.no_grad()
    def _select(self, image, image_embedding=None):
        """Decode every grid prompt and run upstream's filters on numbers only.  -> None (nothing survives) or a dict with the
        candidates' low-resolution logits `low`, the surviving candidates' indices into it in NMS order `idx`, their
        `iou`, `points`, `stability`, `boxes` (device tensors) and the geometry."""
        c = self.cfg
        dec, dev = self.decoder, self.decoder.device
        S = self.encoder.cfg["img_size"] if self.encoder is not None else dec.img_size
        if image_embedding is None:
            st = self.set_image(image)
        else:
            H, W = np.asarray(image).shape[:2]
            st = dict(tokens=dec.image_tokens(image_embedding), emb_hw=tuple(image_embedding.shape[-2:]), orig=(H, W),
                      inp=preprocess_shape(H, W, S))
        (H, W), (in_h, in_w) = st["orig"], st["inp"]
        pts_all = build_point_grid(c["points_per_side"]) * np.array([[W, H]])
        crop_box = [0, 0, W, H]
        # Two phases.  (1) Every prompt is decoded and its three candidates are filtered on numbers only: predicted IoU,
        # then the stability score / box of the mask AT THE ORIGINAL RESOLUTION from a statistics-only post-processing
        # pass (no mask is written: 3072 full-resolution masks per image were 0.8 GB of byte stores).  (2) After the box
        # NMS the masks of the surviving records alone are written, from their kept low-resolution logits -- the same
        # kernel, so the same pixels as a one-pass run.  `points_per_batch` is upstream's memory knob and does not change
        # any result (prompts are independent): the decoder runs `decode_batch` prompts at a time (constructor argument).
        scale = torch.tensor([in_w / W, in_h / H], device=dev)
        step = max(int(c["points_per_batch"]), self.decode_batch) if self.decode_batch >= DECODE_BATCH else self.decode_batch
        lows, ious, ptss = [], [], []
        for s in range(0, len(pts_all), step):
            ck = (H, W, in_h, in_w, int(c["points_per_side"]), s, step)
            if ck not in self._sparse_cache:                 # the grid's prompts depend on the image SIZE only
                if len(self._sparse_cache) > 16:
                    self._sparse_cache.clear()
                p = torch.as_tensor(pts_all[s:s + step], dtype=torch.float32, device=dev)
                self._sparse_cache[ck] = (p, dec.embed_points((p * scale)[:, None, :], torch.ones(len(p), 1)))
            p, sparse = self._sparse_cache[ck]
            decode = dec.predict_masks_graph if self.use_graph and hasattr(dec, "predict_masks_graph") else dec.predict_masks
            low, iou = decode(st["tokens"], st["emb_hw"], sparse, True)
            if len(pts_all) > step and decode is not dec.predict_masks:
                low, iou = low.clone(), iou.clone()          # a replay's outputs are overwritten by the next chunk's replay
            lows.append(low.flatten(0, 1)); ious.append(iou.flatten(0, 1)); ptss.append(p.repeat_interleave(3, dim=0))
        low = lows[0] if len(lows) == 1 else torch.cat(lows)          # the candidates' logits stay where the decoder wrote them:
        iou = ious[0] if len(ious) == 1 else torch.cat(ious)          # every filter below works on an index list
        ppts = ptss[0] if len(ptss) == 1 else torch.cat(ptss)
        idx = torch.nonzero(iou > c["pred_iou_thresh"]).reshape(-1)
        if idx.numel() == 0:
            return None
        geom = ((in_h, in_w), (H, W), S)
        _, stats = ops.sam_mask_postprocess(low, (in_h, in_w), (H, W), S, c["mask_threshold"], c["stability_score_offset"],
                                            want_masks=False, index=idx.int())
        stab = stats[:, 0] / stats[:, 1]
        boxes = torch.where((stats[:, 4:5] < 0), torch.zeros_like(stats[:, 2:6]), stats[:, 2:6])
        k = (stab >= c["stability_score_thresh"]) & ~is_box_near_crop_edge(boxes, crop_box, [0, 0, W, H])
        idx, stab, boxes = idx[k], stab[k], boxes[k]
        if idx.numel() == 0:
            return None
        iou, ppts = iou[idx], ppts[idx]
        order = nms(boxes, iou, c["box_nms_thresh"]).to(dev)
        return dict(low=low, idx=idx[order], iou=iou[order], points=ppts[order], stability=stab[order], boxes=boxes[order], geom=geom)

    def _masks(self, sel, lo, hi):
        """uint8 masks [hi - lo, H, W] of the selected records lo .. hi - 1 (device)."""
        c = self.cfg
        (inp, orig, S) = sel["geom"]
        masks, _ = ops.sam_mask_postprocess(sel["low"], inp, orig, S, c["mask_threshold"], c["stability_score_offset"],
                                            index=sel["idx"][lo:hi].int())
        return masks

    @torch.no_grad()
    def generate(self, image, image_embedding=None):
        sel = self._select(image, image_embedding)
        if sel is None:
            return []
        H, W = sel["geom"][1]
        masks = self._masks(sel, 0, len(sel["idx"])).bool()
        areas = masks.flatten(1).sum(1).cpu().tolist()
        m_np = masks.cpu().numpy()
        iou_l, pts_l, stab_l, box_l = (sel[k].cpu().tolist() for k in ("iou", "points", "stability", "boxes"))
        return [dict(segmentation=m_np[i], area=int(areas[i]), bbox=[box_l[i][0], box_l[i][1], box_l[i][2] - box_l[i][0],
                                                                      box_l[i][3] - box_l[i][1]],
                     predicted_iou=float(iou_l[i]), point_coords=[pts_l[i]], stability_score=float(stab_l[i]),
                     crop_box=[0, 0, W, H]) for i in range(len(iou_l))]

    @torch.no_grad()
    def generate_id_map(self, image, image_embedding=None):
        """What the hot path consumes from `generate` (sam2image.py:117-120: `show_anns(mask_generator.generate(image))`):
        the id map -- record i paints i + 1 over its mask in list order, later records over earlier ones, i.e. the
        LARGEST record number covering a pixel -- built on the device from the same mask bytes `generate` returns, without
        the records' full-size masks ever crossing to the host.  -> (int32 [H, W] device tensor, number of records);
        `host.show_anns_from_id_map` turns it into show_anns' return value."""
        sel = self._select(image, image_embedding)
        H, W = np.asarray(image).shape[:2]
        if sel is None:
            return torch.zeros((H, W), dtype=torch.int32, device=self.decoder.device), 0
        n = len(sel["idx"])
        if W <= ops.SAM_ID_MAP_MAX_W:
            # round 6: one launch walks the records per pixel from the last to the first covering one (ops.sam_id_map: the mask
            # kernel's arithmetic, the same bit per record and pixel) -- no full-resolution mask is written or reduced
            (inp, orig, S) = sel["geom"]
            return ops.sam_id_map(sel["low"], inp, orig, S, self.cfg["mask_threshold"], index=sel["idx"].int()), n
        idm = torch.zeros((H, W), dtype=torch.int16, device=self.decoder.device)
        for lo in range(0, n, 512):
            hi = min(n, lo + 512)
            m = self._masks(sel, lo, hi).ne(0)
            ids = torch.arange(lo + 1, hi + 1, dtype=torch.int16, device=idm.device)
            idm = torch.maximum(idm, (m * ids[:, None, None]).amax(0))
        return idm.int(), n


def remove_small_regions(mask, area_thresh, mode):
    """segment_anything utils/amg.py remove_small_regions (reference call: sam2groundingdino_edit.py:186-188, "holes",
    400): connected components with 8-connectivity (scipy.ndimage.label in place of cv2.connectedComponentsWithStats --
    the same components); "holes" fills background holes smaller than area_thresh, "islands" drops foreground islands
    smaller than it (the largest one survives if all are small).  -> (bool mask, changed)."""
    from scipy import ndimage
    if mode not in ("holes", "islands"):
        raise AssertionError(mode)
    holes = mode == "holes"
    working = (holes ^ np.asarray(mask).astype(bool)).astype(np.uint8)
    regions, n = ndimage.label(working, structure=np.ones((3, 3), int))
    sizes = np.bincount(regions.ravel(), minlength=n + 1)[1:]
    small = [i + 1 for i, sz in enumerate(sizes) if sz < area_thresh]
    if not small:
        return np.asarray(mask).astype(bool), False
    fill = [0] + small
    if not holes:
        fill = [i for i in range(n + 1) if i not in fill]
        if not fill:
            fill = [int(np.argmax(sizes)) + 1]
    return np.isin(regions, fill), True


class _ResizeLongestSide:
    """The `sam_predictor.transform` the reference reaches into (sam2groundingdino_edit.py:177)."""

    def __init__(self, target_length):
        self.target_length = target_length

    def apply_boxes_torch(self, boxes, original_size):
        h, w = original_size
        nh, nw = preprocess_shape(h, w, self.target_length)
        b = boxes.float().reshape(-1, 2, 2).clone()
        b[..., 0] *= nw / w
        b[..., 1] *= nh / h
        return b.reshape(-1, 4)

    def apply_coords_torch(self, coords, original_size):
        h, w = original_size
        nh, nw = preprocess_shape(h, w, self.target_length)
        c = coords.float().clone()
        c[..., 0] *= nw / w
        c[..., 1] *= nh / h
        return c


class SamPredictor:
    """`SamPredictor(sam)`: click prompts (editany_lora.py:527-543: set_image + predict(point_coords, point_labels)) and
    box prompts (sam2groundingdino_edit.py:176-183: transform.apply_boxes_torch + predict_torch(boxes=...))."""

    def __init__(self, encoder, decoder):
        self._gen = SamAutomaticMaskGenerator(encoder, decoder)
        self.decoder = decoder
        self.transform = _ResizeLongestSide(decoder.img_size)

    @torch.no_grad()
    def predict_torch(self, point_coords=None, point_labels=None, boxes=None, mask_input=None, multimask_output=True,
                      return_logits=False):
        """Prompts ALREADY in the input frame (`transform.apply_*`), batched: point_coords [B, N, 2] + point_labels
        [B, N] and / or boxes [B, 4] -> (masks [B, C, H, W] bool or logits, iou [B, C], low-res logits [B, C, h, w])."""
        if mask_input is not None:
            raise NotImplementedError("mask prompts (mask_downscaling) are outside the path: no reference caller passes one")
        st, dec = self._st, self.decoder
        (H, W), (in_h, in_w) = st["orig"], st["inp"]
        sparse = None
        if point_coords is not None:
            if point_labels is None:
                raise AssertionError("point_labels must be supplied if point_coords is supplied.")
            sparse = dec.embed_points(point_coords, point_labels) if boxes is None else \
                dec.embed_points(point_coords, point_labels)[:, :-1]          # the padding point only without boxes
        if boxes is not None:
            be = dec.embed_boxes(boxes)
            sparse = be if sparse is None else torch.cat([sparse, be], dim=1)
        if sparse is None:
            raise ValueError("predict_torch needs point or box prompts")
        low, iou = dec.predict_masks(st["tokens"], st["emb_hw"], sparse, multimask_output)
        masks = postprocess_masks(low, (in_h, in_w), (H, W), dec.img_size)
        if not return_logits:
            masks = masks > 0.0
        return masks, iou, low

    def set_image(self, image_u8_hwc):
        self._st = self._gen.set_image(image_u8_hwc)

    @torch.no_grad()
    def predict(self, point_coords, point_labels, multimask_output=True, return_logits=False):
        st, dec = self._st, self.decoder
        (H, W), (in_h, in_w) = st["orig"], st["inp"]
        p = torch.as_tensor(np.asarray(point_coords), dtype=torch.float32, device=dec.device)
        p = p * torch.tensor([in_w / W, in_h / H], device=dec.device)
        lab = torch.as_tensor(np.asarray(point_labels), dtype=torch.float32)
        sparse = dec.embed_points(p[None], lab[None])
        low, iou = dec.predict_masks(st["tokens"], st["emb_hw"], sparse, multimask_output)
        masks = postprocess_masks(low, (in_h, in_w), (H, W), dec.img_size)
        if not return_logits:
            masks = masks > 0.0
        return masks[0].cpu().numpy(), iou[0].cpu().numpy(), low[0].cpu().numpy()
