"""Evidence for the mAP half of the metric without datasets: train the detector from the SAME seed in
fp32 and in the bench precision (bf16 MFMA operands) on synthetic scenes whose classes are learnable from shape, through
cagroup3d_amd/train.py's loop (AdamW, step decay, clip 10, semantic-threshold schedule), evaluate both on HELD-OUT scenes
with indoor_eval, and report the loss curves and mAP@0.25 / 0.50.  GPU only.

    python tools/synthetic_convergence.py [--scenes 40 --val 8 --epochs 30 --config S50k-shape] > profiles/r02_synthetic_convergence.json
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OMP_NUM_THREADS", "1")
import numpy as np
import torch
from cagroup3d_amd import build_model, me, train


# precision legs: operand type of the backbone's convolutions, arithmetic of the two heads, storage of the backbone's rows
#   fp32            fp32 everywhere (the reference's arithmetic)
#   bench           bf16 backbone (activations / gradients stored as bf16) + split heads (fp32-accurate: three bf16 passes) -- bench.py's default
#   bf16-fp32rows   the same with the backbone's rows stored as fp32 (the arithmetic of rounds 1-4)
#   bf16            bf16 operands in the heads too (rows as in `bench`)
#   bf16-backbone   bf16 backbone, fp32 MFMA operands in the heads
LEGS = {"fp32": (0, None, True), "bench": (1, me.PREC_SPLIT, True), "bf16-fp32rows": (1, me.PREC_SPLIT, False),
        "bf16": (1, None, True), "bf16-backbone": (1, 0, True)}


def run(precision, args):
    from cagroup3d_amd import engine
    me.PRECISION, me.HEAD_PRECISION, engine.ACT_BF16 = LEGS[precision]
    me._WeightPlan.reset()
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    model, cfg = build_model.build_cagroup3d("scannet", seed=args.seed)
    model = model.cuda()
    oc = cfg.OPTIMIZATION
    oc["DECAY_STEP_LIST"] = [int(args.epochs * 0.7), int(args.epochs * 0.9)]     # the reference's 7 / 9 of 10 epochs
    ds = train.SyntheticIndoorDataset(args.config, args.scenes, args.batch)
    val = train.SyntheticIndoorDataset(args.config, args.val, args.batch, first_scene=100000)      # held out
    opt = train.build_optimizer(model, oc)
    sched = train.build_scheduler(opt, len(ds), oc)
    losses, it, t0 = [], 0, time.time()
    quiet = lambda *a, **k: None
    for epoch in range(args.epochs):
        print("[%s] epoch %d it %d %.0fs" % (precision, epoch, it, time.time() - t0), file=sys.stderr, flush=True)
        it = train.train_one_epoch(model, opt, sched, ds, min(epoch, args.thr_epochs), it, oc.GRAD_NORM_CLIP, log=quiet, losses=losses)
    torch.cuda.synchronize()
    secs = time.time() - t0
    print("[%s] eval" % precision, file=sys.stderr, flush=True)
    res = train.eval_one_epoch(model, val, cfg.CLASS_NAMES, torch.device("cuda"), log=quiet)
    per = len(ds)
    curve = [float(np.mean(losses[e * per:(e + 1) * per])) for e in range(args.epochs)]
    return {"precision": precision, "iterations": it, "train_seconds": secs, "epoch_mean_loss": curve,
            "first_losses": losses[:5], "last_losses": losses[-5:],
            "mAP_0.25": float(res["mAP_0.25"]), "mAP_0.50": float(res["mAP_0.50"]),
            "mAR_0.25": float(res["mAR_0.25"]), "mAR_0.50": float(res["mAR_0.50"])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="S50k-shape")
    ap.add_argument("--scenes", type=int, default=40)
    ap.add_argument("--val", type=int, default=8)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--epochs", type=int, default=30)
    ap.add_argument("--thr-epochs", type=int, default=5, help="epochs over which the semantic threshold falls (0.15 -> 0.05), as in the 10-epoch recipe")
    ap.add_argument("--seed", type=int, default=0, help="seed of the weights' initialisation and of the host RNG streams")
    ap.add_argument("--repeat-fp32", action="store_true", help="a second fp32 run from the same seed: the run-to-run noise (fp32 atomics)")
    args = ap.parse_args()
    out = {"what": "CAGroup3D trained from seed " + str(args.seed) + " on %d synthetic %s scenes (classes learnable from shape), %d epochs x %d iterations, "
                   "batch %d, AdamW 1e-3, decay x0.1 at 70%% / 90%%, clip 10; evaluated on %d held-out scenes with indoor_eval"
                   % (args.scenes, args.config, args.epochs, -(-args.scenes // args.batch), args.batch, args.val),
           "runs": [run(p, args) for p in os.environ.get("CG3D_CONV_RUNS", "fp32,bench,bf16-fp32rows,bf16").split(",")]}
    if args.repeat_fp32:
        out["runs"].append(dict(run("fp32", args), precision="fp32 (second run, same seed)"))
    a, b = out["runs"][0], out["runs"][min(1, len(out["runs"]) - 1)]
    out["bf16_minus_fp32"] = {"final_epoch_loss": b["epoch_mean_loss"][-1] - a["epoch_mean_loss"][-1],
                              "mAP_0.25": b["mAP_0.25"] - a["mAP_0.25"], "mAP_0.50": b["mAP_0.50"] - a["mAP_0.50"]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
