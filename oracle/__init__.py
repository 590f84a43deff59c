"""ORACLE PACKAGE -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference algorithm for the hot path (SAM ViT image encoder -> ControlNet +
UNet DDIM loop -> VAE).  Importers allowed: tests/, __graft_entry__.smoke(), bench.py `cpu_baseline`.
The product package (editanything_amd/) never imports it and has no CPU fallback.

Parity status:
  * LDM half (ldm_oracle.py): PINNED -- checked against the reference's own modules imported from
    /root/reference (oracle/ref_import.py) and against the golden vectors they produced
    (tests/golden/ldm_*.npz, generator: oracle/make_golden.py).
  * SAM half (sam_oracle.py): segment_anything is a third-party, un-vendored, un-pinned dependency
    (absent from /root/reference); the restatement follows its published algorithm and is pinned
    against the independent HF port installed in this image (transformers.models.sam
    SamVisionEncoder) -- golden vectors tests/golden/sam_*.npz.
  * Host integer logic (host_oracle.py): restated from sam2image.py / annotator/util.py; bit-exact.
"""
