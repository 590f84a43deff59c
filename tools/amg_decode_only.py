"""One image's worth of SAM prompt decoding (1024 grid prompts, ViT-H sized embedding, random weights), repeated: the launch
list rocprofv3 sees is the decoder's alone (tools/gpu_visit.sh profpy:tools/amg_decode_only.py -> profiles/r03_amg_*)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editanything_amd import amg, arch, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
ops.workspace(dev)
dec = amg.SamPromptDecoder(synth.synth_state_dict_torch(arch.sam_decoder_param_shapes(), 12), dev)
emb = torch.randn(1, 256, 64, 64, generator=torch.Generator().manual_seed(0)).to(dev)
tokens = dec.image_tokens(emb)
pts = torch.as_tensor(amg.build_point_grid(32) * 1024.0, dtype=torch.float32, device=dev)
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for _ in range(REPS + 1):
    sparse = dec.embed_points(pts[:, None, :], torch.ones(len(pts), 1))
    low, iou = dec.predict_masks(tokens, (64, 64), sparse, True)
torch.cuda.synchronize()
print("decoded", tuple(low.shape), "x", REPS + 1)
