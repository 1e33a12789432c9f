"""`python -m robigo_luculenta_amd` -- the reference's main.rs (main.rs:45-67) on one MI355X: renders the
built-in scene with the App worker pool and writes output.ppm (+ buffer.raw) when the batch budget is done."""
import argparse

from . import SCENE_DEMO, SCENE_GLASS_STRESS, app_run


def main():
    ap = argparse.ArgumentParser(prog="python -m robigo_luculenta_amd")
    ap.add_argument("--width", type=int, default=1280)    # main.rs:47
    ap.add_argument("--height", type=int, default=720)    # main.rs:48
    ap.add_argument("--batches", type=int, default=32, help="trace tasks to run (the reference runs forever)")
    ap.add_argument("--photons-per-batch", type=int, default=64 * 1024 * 512,
                    help="paths per trace task; the reference's 524288 (trace_unit.rs:67) keeps an MI355X busy for "
                         "0.2 ms only, so the default is 64 of those")
    ap.add_argument("--concurrency", type=int, default=2)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--scene", choices=["demo", "glass"], default="demo")
    ap.add_argument("--unfused", action="store_true", help="keep the reference's separate Trace and Plot work")
    ap.add_argument("--output", default="output.png", help="*.png or *.ppm")  # main.rs:61
    ap.add_argument("--checkpoint", default=None, help="buffer.raw to write (and to resume from with --resume)")
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args()
    print("rendering %d batches at %dx%d" % (a.batches, a.width, a.height))
    rgb, st = app_run(a.width, a.height, a.batches, concurrency=a.concurrency, device=a.device, seed=a.seed,
                      photons_per_batch=a.photons_per_batch,
                      scene=SCENE_DEMO if a.scene == "demo" else SCENE_GLASS_STRESS, fused=not a.unfused,
                      output_ppm=a.output, checkpoint=a.checkpoint, resume=a.resume, verbose=not a.quiet)
    print("%d trace tasks, %.1f Mpaths, %.1f Mrays in %.2f s (%.1f Mrays/s, %.1f reference batches/sec); wrote %s"
          % (st["batches"], st["paths"] / 1e6, st["segments"] / 1e6, st["seconds"], st["segments"] / st["seconds"] / 1e6,
             st["paths"] / 524288.0 / st["seconds"], a.output))


if __name__ == "__main__":
    main()
