"""BASELINE config 3 at length, as Tracking sees it: the per-frame sequence of tools/streamed_frontend.cpp over 3 682
frames — the length and the 20 Hz time stamps of EuRoC MH_01 (Examples/Monocular/EuRoC_TimeStamps/MH01.txt, mono_euroc.cc:84-160) — on the
S-EuRoC-640 image set walked forth and back, through the drop-in (liborbx.so) back to back AND paced, and through the reference-compiled
build; digests over ALL frames must be equal.  Run on the GPU box:  python tools/config3_full.py [--frames 3682] [--images 256] > profiles/config3_full_rN.txt"""
import argparse
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import world_util as wu  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=3682)
    ap.add_argument("--images", type=int, default=256)
    ap.add_argument("--no-ref", action="store_true")
    ap.add_argument("--no-paced", action="store_true")
    ap.add_argument("--spin", action="store_true", help="a third drop-in run paced by busy-waiting instead of sleeping (is the paced tail the CPU's or the GPU's idle state?)")
    ap.add_argument("--env", action="append", default=[], help="KEY=VALUE for the drop-in runs (e.g. ORBX_KEEP_WARM=1)")
    a = ap.parse_args()
    from orb_slam3_modified_amd import build
    for kv in a.env:
        k, v = kv.split("=", 1)
        os.environ[k] = v
    td = tempfile.mkdtemp()
    raw, voc = wu.frontend_inputs(td, a.images, 480, 640, 1000)
    stamps = wu.mh01_stamps(td)
    exe = wu.build_frontend("orbx")
    print(f"# config 3 at length: {a.frames} frames, {a.images} S-EuRoC-640 images forth and back, 1000 features, 8 levels; {build.stamp()}")
    if a.env:
        print(f"# environment of the drop-in runs: {a.env}")
    rows = {}
    rows["dropin_back_to_back"] = wu.run_frontend(exe, raw, 480, 640, a.images, 1000, voc, 1, timeout=1200, frames=a.frames)
    if not a.no_paced:
        rows["dropin_paced_20hz"] = wu.run_frontend(exe, raw, 480, 640, a.images, 1000, voc, 1, timeout=1200, frames=a.frames, stamps=stamps, pace=1)
    if a.spin:
        rows["dropin_paced_20hz_busy_wait"] = wu.run_frontend(exe, raw, 480, 640, a.images, 1000, voc, 1, timeout=1200, frames=a.frames, stamps=stamps, pace=2)
    if not a.no_ref and os.path.exists(wu.REF_FRONTEND_EXE):
        rows["reference_back_to_back"] = wu.run_frontend(wu.REF_FRONTEND_EXE, raw, 480, 640, a.images, 1000, voc, 1, timeout=2400, frames=a.frames)
    digests = {k: v["results_digest"] for k, v in rows.items()}
    for k, v in rows.items():
        print(f"\n## {k}: {v['build']}")
        print(f"frames timed {v['frames_timed']}, features/frame {v['features_per_frame']}, matches last/local {v['matches_last_per_frame']} / "
              f"{v['matches_local_per_frame']}, wide retries {v['wide_retries']}, digest {v['results_digest']}, stream {json.dumps(v['stream'])}")
        print("| call | mean | p50 | p90 | p99 | max | (ms)")
        print("|---|---:|---:|---:|---:|---:|")
        for c, q in v["percentiles"].items():
            print(f"| {c} | {q['mean']:.4f} | {q['p50']:.4f} | {q['p90']:.4f} | {q['p99']:.4f} | {q['max']:.4f} |")
    same = len(set(digests.values())) == 1
    print(f"\ndigest over all {a.frames} frames equal across {sorted(digests)}: {same}")
    if "dropin_paced_20hz" in rows:
        r = rows["dropin_paced_20hz"]["percentiles"]["four_calls_ms"]["p50"] / rows["dropin_back_to_back"]["percentiles"]["four_calls_ms"]["p50"]
        print(f"paced p50 / back-to-back p50 of the four calls: {r:.3f}")
    print(json.dumps(rows))
    sys.exit(0 if same else 1)


if __name__ == "__main__":
    main()
