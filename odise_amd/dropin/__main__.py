from . import OVERLAY_DIR

print(OVERLAY_DIR)
