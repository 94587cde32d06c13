import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One libodise_hip context for the whole GPU test session (fails loudly if the .so is missing)."""
    import os
    from odise_amd.runtime import Context
    c = Context(0)
    if os.environ.get("ODISE_TEST_MASKCLIP_PASSES"):   # developer aid: which form of MaskCLIP a difference comes from (include/odise_hip.h ODISE_OPT_MASKCLIP_PASSES)
        c.set_option(c.OPT_MASKCLIP_PASSES, int(os.environ["ODISE_TEST_MASKCLIP_PASSES"]))
    yield c
    c.close()


_FULLSIZE = {}


@pytest.fixture(scope="module")
def fullsize_model(ctx):
    """ONE full-size device model (SD-v1 UNet 859.5 M, AutoencoderKL, CLIP ViT-L/14@336, ODISE heads; weights of tests/fullsize.py) shared by
    tests/test_gpu_fullsize.py and tests/test_gpu_fullsize_1280.py: packing and uploading 1.28 G parameters takes longer than most of the
    tests.  A context holds one model, and other test modules load theirs into the same session context: the model is rebuilt whenever it is
    no longer the resident one (`ctx.model_owner`).  The module fixtures load their own category head (the null embedding follows the test
    image) and vocabulary on top."""
    hip = _FULLSIZE.get("hip")
    if hip is None or ctx.model_owner is not hip:
        from fullsize import build_models, export_state, reference
        from odise_amd.pipeline import HipCategoryODISE
        ext, bb, head = build_models()
        _, heads, _ = reference(bb, head, ext, 1024, 133, 254)
        hip = HipCategoryODISE(ctx, export_state(ext, bb, head, heads), overlap_threshold=0.8)
        _FULLSIZE["hip"] = hip
    return hip


GOLDEN = os.path.join(ROOT, "tests", "golden")
