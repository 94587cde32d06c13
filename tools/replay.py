"""Replays the reference's two calling conventions around the model on the device path (SURVEY.md 8d config 1 and timing convention):

  demo   `VisualizationDemo.predict` (demo/demo.py:153-171) with demo.py's default vocabulary shape (`--label COCO ADE LVIS`,
         demo.py:296-362: 133 + 150 + 1203 = 1486 classes / 2482 prompt strings, `overlap_threshold=0`, alpha / beta = 0.35 / 0.65,
         demo.py:316-318): one 512x512x3 uint8 picture -> ResizeShortestEdge(1024, 2560) (Pillow-identical bilinear on the device) ->
         float32 CHW `{"image", "height", "width"}` -> `model([inputs])[0]`; asserts the reference's output format.
  eval   `inference_on_dataset` (odise/evaluation/evaluator.py:60-142): batches from a loader, `min(5, total - 1)` warm-up iterations
         excluded, device synchronised after every batch, wall clock by `time.perf_counter`; prints s/iter and images/s.

  overlay  the same evaluator loop through the DROP-IN model: the overlay's `CategoryODISE` (odise_amd/dropin/zoo.py, the released label model)
         moved to the device, wrapped in the reference-protocol `OpenPanopticInference` stand-in, batches of CPU uint8 CHW tensors as
         detectron2's DatasetMapper yields them; the results are device tensors the library wrote in place.  Compared with `eval` (the
         ctypes-level model on resident uint8 pictures) this prices the boundary: picture upload + torch allocation of the outputs.

Weights: random tensors of the real shapes (odise_amd/synthetic.py); text banks: seeded random rows of the real count.
usage: replay.py demo | eval | overlay [--iters N] [--batch B]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (the seeded box-filtered pictures of SURVEY 8d)
from odise_amd.ingest import resize_shortest_edge_shape  # noqa: E402


def demo_predict(hip, ctx, picture_u8_hwc):
    """demo.py:153-171.  `picture_u8_hwc`: np.uint8 [H,W,3]."""
    height, width = picture_u8_hwc.shape[:2]
    nh, nw = resize_shortest_edge_shape(height, width, 1024, 2560)                                   # T.ResizeShortestEdge(1024, 2560)
    resized = ctx.resize_bilinear_u8(ctx.to_device(picture_u8_hwc), nh, nw).numpy()
    image = resized.astype("float32").transpose(2, 0, 1)                                               # torch.as_tensor(image.astype("float32").transpose(2, 0, 1))
    inputs = {"image": image, "height": height, "width": width}
    return hip([inputs])[0], (nh, nw)


def check_demo_outputs(pred, K, height, width):
    sem, (pan, info), inst = pred["sem_seg"], pred["panoptic_seg"], pred["instances"]
    assert sem.shape == (K, height, width) and sem.dtype == np.float32
    assert pan.shape == (height, width) and pan.dtype == np.int32
    assert all(set(s) == {"id", "isthing", "category_id"} and 0 <= s["category_id"] < K for s in info)
    assert sorted(s["id"] for s in info) == list(range(1, len(info) + 1)) and set(np.unique(pan)) <= set([0] + [s["id"] for s in info])
    n = len(inst["scores"])
    assert inst["pred_masks"].shape == (n, height, width) and inst["pred_classes"].shape == (n,) and inst["pred_classes"].dtype == np.int64
    return len(info), n


def inference_on_dataset(hip, ctx, loader, total):
    """evaluator.py:60-142 without an evaluator: returns (seconds per iteration, images per second) over the iterations after warm-up."""
    num_warmup = min(5, total - 1)
    start_time = time.perf_counter()
    total_compute_time, images = 0.0, 0
    for idx, inputs in enumerate(loader):
        if idx == num_warmup:
            start_time = time.perf_counter()
            total_compute_time, images = 0.0, 0
        start_compute_time = time.perf_counter()
        hip.forward(inputs, to_host=False)                 # outputs stay on the device, as the reference's are device tensors
        ctx.sync()                                         # torch.cuda.synchronize()
        total_compute_time += time.perf_counter() - start_compute_time
        images += len(inputs)
    iters = total - num_warmup
    return total_compute_time / iters, (time.perf_counter() - start_time) / iters, images / (time.perf_counter() - start_time)


class _Banks:
    """Stands in for tokenizer + CLIP text tower (no BPE merges file here): prompt strings -> rows of seeded banks."""

    def __init__(self, table):
        self.table = table

    def tokenize(self, strings):
        return list(strings)

    def build_text_embed(self, strings):
        return np.stack([self.table[s] for s in strings]).astype(np.float32)


def overlay_eval(ctx, args):
    import torch
    from odise_amd import dropin
    from odise_amd.dropin import zoo
    from odise_amd.synthetic import synthetic_state, synthetic_vocabulary
    K, K_TOT = 133, 254
    cat, clp, sizes, overlap = synthetic_vocabulary(K, K_TOT, 768)
    labels, table, row = [], {}, 0
    for k, n in enumerate(sizes):
        names = [f"class{k}_{j}" for j in range(int(n))]
        labels.append(names)
        for s in names:
            table[s], table[f"a photo of a {s}."] = cat[row], clp[row]
            row += 1
    dropin.set_context(ctx)
    dropin.set_text_tools(_Banks(table).tokenize, _Banks(table))
    model = zoo.category_odise_with_label(labels, range(80), [bool(o) for o in overlap])
    zoo.load_flat_state(model, synthetic_state())
    model.eval().to("cuda")
    model.load_open_state_dict({k: (labels if k.endswith("test_labels") else v) for k, v in model.open_state_dict().items()})
    batches = [[{"image": torch.from_numpy(np.ascontiguousarray(bench.image_u8(1024, (i * args.batch + b) % 8).transpose(2, 0, 1))), "height": 1024, "width": 1024}
                for b in range(args.batch)] for i in range(args.iters)]
    total, num_warmup = len(batches), min(5, len(batches) - 1)
    t_comp, images = 0.0, 0
    with torch.no_grad():
        for idx, inputs in enumerate(batches):
            if idx == num_warmup:
                start, t_comp, images = time.perf_counter(), 0.0, 0
            t0 = time.perf_counter()
            out = model(inputs)
            torch.cuda.synchronize()
            t_comp += time.perf_counter() - t0
            images += len(inputs)
    assert out[0]["sem_seg"].is_cuda and out[0]["panoptic_seg"][0].is_cuda and out[0]["instances"].pred_masks.is_cuda
    iters = total - num_warmup
    wall = time.perf_counter() - start
    print(f"eval-loop replay THROUGH THE DROP-IN MODEL (overlay CategoryODISE on cuda, CPU uint8 CHW pictures in, device tensors out; evaluator.py "
          f"convention): {t_comp / iters:.4f} s/iter inference, {wall / iters:.4f} s/iter total, {images / wall:.2f} images/s at batch {args.batch} x 1024x1024")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["demo", "eval", "overlay"])
    ap.add_argument("--iters", type=int, default=15)
    ap.add_argument("--batch", type=int, default=4)
    args = ap.parse_args()
    if args.mode == "overlay":
        import torch   # before the library: the process then runs ONE HIP runtime (torch's), which both torch and libodise_hip.so bind to
        torch.cuda.init()
    from odise_amd.pipeline import HipCategoryODISE
    from odise_amd.runtime import Context
    from odise_amd.synthetic import synthetic_state, synthetic_vocabulary
    ctx = Context(0)
    if args.mode == "overlay":
        return overlay_eval(ctx, args)
    if args.mode == "demo":
        K, K_TOT, things = 1486, 2482, 80 + 100 + 1203                                             # COCO things + ADE things + all of LVIS
        hip = HipCategoryODISE(ctx, synthetic_state(), overlap_threshold=0.0)
        cat, clp, sizes, overlap = synthetic_vocabulary(K, K_TOT, 768)
        hip.set_vocabulary(cat, clp, sizes, overlap, set(range(things)), 0.35, 0.65)
        pic = bench.image_u8(512, 0)
        t0 = time.perf_counter()
        pred, net_hw = demo_predict(hip, ctx, pic)
        dt = time.perf_counter() - t0
        segs, n = check_demo_outputs(pred, K, 512, 512)
        print(f"demo replay: 512x512 picture -> network input {net_hw} (4 crops), K = {K} classes / {K_TOT} strings; outputs sem_seg {pred['sem_seg'].shape}, "
              f"{segs} panoptic segments, {n} instances; first call {dt * 1e3:.0f} ms (includes the read-back of {pred['sem_seg'].nbytes / 1e9:.2f} GB of sem_seg)")
    else:
        hip = HipCategoryODISE(ctx, synthetic_state(), overlap_threshold=0.8)
        cat, clp, sizes, overlap = synthetic_vocabulary(133, 254, 768)
        hip.set_vocabulary(cat, clp, sizes, overlap, set(range(80)), 0.3, 0.7)
        batches = [[{"image": ctx.to_device(bench.image_u8(1024, (i * args.batch + b) % 8)), "height": 1024, "width": 1024} for b in range(args.batch)]
                   for i in range(args.iters)]
        comp, tot, ips = inference_on_dataset(hip, ctx, batches, len(batches))
        print(f"eval-loop replay (evaluator.py convention, 5 warm-up iterations excluded, sync per batch): {comp:.4f} s/iter inference, {tot:.4f} s/iter total, "
              f"{ips:.2f} images/s at batch {args.batch} x 1024x1024")


if __name__ == "__main__":
    main()
