"""CPU oracle for the ODISE inference hot path — TEST INFRASTRUCTURE ONLY.

Everything under oracle/ restates the reference's algorithm on the CPU (plain torch fp32 / numpy), citing
the reference file:line each function follows.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import it; the product path (odise_amd/) never does.

Pinning status:
  * oracle.msda         — PINNED: checked against the reference's own `ms_deform_attn_core_pytorch`
                          (imported from /root/reference in the build container by
                          tests/golden/make_golden.py; vectors committed under tests/golden/).
  * oracle.m2f / oracle.odise_model — PINNED: replayed against golden vectors written by the reference's own
                          `MaskFormerHead` / `MSDeformAttnPixelDecoder` / `ODISEMultiScaleMaskedTransformerDecoder`
                          and by its `CategoryODISE.forward` eval branch (tests/golden/make_golden_m2f.py,
                          make_golden_heads.py import them from /root/reference with the third-party imports
                          stubbed by tests/golden/ref_stubs.py; tests/test_oracle_golden.py).
  * oracle.backbone / oracle.ldm_extractor — DRIVER LOGIC PINNED the same way (make_golden_backbone.py: the reference's
                          FeatureExtractorBackbone over a stand-in tap extractor; make_golden_extractor.py: the reference's
                          LdmImplicitCaptionerExtractor / LdmExtractor forward and its own GaussianDiffusion.q_sample walking
                          the oracle's UNet / VAE / CLIP modules).
  * oracle.jpeg / oracle.eval_ops — PINNED against Pillow (= libjpeg-turbo / Resample.c), the library behind
                          the reference's `read_image` and `ResizeTransform`.
  * oracle.sd_unet / sd_vae / clip_vit / d2_blocks — PARITY UNPINNED: the arithmetic lives in pip
                          dependencies that are absent from /root/reference (stable-diffusion-sdkit==2.1.3,
                          open-clip-torch==2.0.2, detectron2 v0.6); restated from their published
                          architectures and anchored on ODISE's call sites.  CLIP is cross-checked against
                          the independent HF `transformers` implementation that is installed here.
"""
