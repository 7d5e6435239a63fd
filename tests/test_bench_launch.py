"""bench.py's N>1 plumbing: `python bench.py --gpus N` with no launcher around it starts its own ranks
(torch.distributed.run on 127.0.0.1), they rendezvous, barrier and MAX-reduce, rank 0 prints ONE JSON line.
CPU box: the GPU-free `--rendezvous-only` form over gloo.  GPU box: a 2-rank dry run of the real bench on ONE device
(`--backend gloo --share-device`), with the configs[2] gather of the packed records inside the timed region."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, timeout):
    """ONE attempt (round 4 retried once: a failure seen "once in about ten runs" whose message was never kept.  Round 5 hunted it
    with tools/launch_flake_hunt.py -- consecutive launches of exactly these commands, everything a failing attempt prints kept:
    30 / 30 of the 8-rank rendezvous on the CPU box and 20 / 20 + 100 / 100 of the 8-rank dry run on MI355X boxes passed,
    profiles/r05_a_launch_flake_hunt_*.json -- so the retry is gone and a failure shows its stderr here)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, BENCH] + args, env=env, cwd=ROOT, timeout=timeout, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True)
    assert out.returncode == 0, "bench.py %s failed (rc %d):\n%s" % (" ".join(args), out.returncode, out.stderr[-6000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]            # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks_and_they_rendezvous():
    line = _run(["--gpus", "2", "--backend", "gloo", "--rendezvous-only"], 300)
    assert line == {"rendezvous": "ok", "n_gpus": 2, "backend": "gloo"}


def test_bench_self_launches_eight_ranks_and_they_rendezvous():
    """The driver's 8-GPU launch shape (one rank per GPU of one node), GPU-free: rendezvous, barrier, MAX over ranks."""
    line = _run(["--gpus", "8", "--backend", "gloo", "--rendezvous-only"], 600)
    assert line == {"rendezvous": "ok", "n_gpus": 8, "backend": "gloo"}


def test_bench_refuses_a_mismatched_launcher_environment():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-only"], env=env, cwd=ROOT, timeout=120,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert out.returncode != 0 and "WORLD_SIZE=3" in out.stderr


@pytest.mark.gpu
def test_bench_two_rank_dry_run_on_one_device():
    line = _run(["--gpus", "2", "--backend", "gloo", "--share-device", "--steps", "40", "--warmup", "8", "--worlds", "2048", "--reps", "5",
                 "--no-cpu-baseline", "--no-full-loop", "--no-configs3", "--no-pmc"], 900)
    assert line["n_gpus"] == 2 and line["steps"] == 40 and line["scaling"] == "weak"
    assert line["value"] > 0 and abs(line["value"] - 2 * 2048 * 4 * 40 / (line["ms_per_step"] * 40e-3)) / line["value"] < 1e-6
    # N > 1: the gather of the packed records is INSIDE the timed region (BASELINE configs[2]); the shard-only rate is the extra
    g = line["extra"]["configs2_gather"]
    assert "error" not in g, g
    assert g["bytes_received_per_rank_per_step"] == 1 * 2048 * 4 * 29 * 4 and g["steps_per_launch"] == 40
    assert g["agent_steps_per_s_with_gather"] == line["value"] and g["agent_steps_per_s_shard_only"] > 0
    assert "gather" in line["config"]["parallelism"]
    r = line["roofline"]
    # roofline.frac IS the SURVEY 8d contract fraction (192 B per agent-step); the bytes the form really moves are frac_moved
    assert r["bound"] in ("latency", "valu-issue", "hbm") and 0 < r["frac_moved"] < r["frac"] == r["frac_contract"]
    assert abs(r["achieved"] - 192 * 2048 * 4 * r["steps_per_launch"] / (r["kernel_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
    assert r["one_step_launch"]["steps_per_launch"] == 1 and 0 < r["one_step_launch"]["frac"]
    # the three hand-over forms, each with its link figures; value is the every-rank form
    assert g["value_is"] == "all" and set(g["forms"]) == {"all", "root", "none"}
    rec = 2048 * 4 * 29 * 4
    for mode, f in g["forms"].items():
        assert f["agent_steps_per_s"] > 0 and f["bytes_per_link_per_step"] == (0 if mode == "none" else rec)
        assert (f["xgmi_frac"] is None) if mode == "none" else (abs(f["link_bound_us_per_step"] - rec / 153e3) < 1e-9 and f["xgmi_frac"] > 0)
    assert g["forms"]["all"]["agent_steps_per_s"] == line["value"]
    # ... and the TOP LEVEL says which of them `value` is, what the wire allows and which curve the scaling target is judged on (round 6):
    # an every-step hand-over of all observations is xGMI-bound below ONE GPU's shard-only rate, which must not read as a scaling failure
    assert line["value_is"] == "all" and line["value_gather_all"] == line["value"]
    assert line["value_shard_only"] == g["forms"]["none"]["agent_steps_per_s"] and line["value_gather_root"] == g["forms"]["root"]["agent_steps_per_s"]
    assert abs(line["wire_bound_agent_steps_per_s"] - 2 * 2048 * 4 / (rec / 153e9)) < 1e-6 * line["wire_bound_agent_steps_per_s"]
    assert line["scaling_judged_on"].startswith("value_shard_only") and "value_shard_only" in line["config"]["parallelism"]
    assert line["extra"]["evidence"]["ran"] == []           # (N > 1: no rocprofv3 child passes, no all-cores baselines)
    # the headline restarts worlds with FRESH generator scenarios (the reference's reset semantics) from the look-ahead rings; the
    # pooled and the in-step sources, and GEN v2, are measured beside it
    assert line["config"]["scenarios"] == "lookahead" and "FRESH generator scenario" in line["config"]["workload"]
    src = line["extra"]["scenario_sources"]
    assert src["headline_source"] == "lookahead"
    for label in ("gen_v1_ring_pool", "gen_v1_ring_instep", "gen_v2_box_lookahead", "gen_v2_box_instep", "training_mix_orca_agents_pool"):
        f = src[label]
        assert "error" not in f and f["value"] > 0 and f["restarts_in_timed_region"] > 0, f
    assert line["timing"]["restarts_in_timed_region"] > 0 and line["timing"]["ms_per_step_mean"] > 0


@pytest.mark.gpu
def test_bench_eight_rank_dry_run_on_one_device():
    """BASELINE configs[2]'s launch shape -- 8 ranks, every rank its shard of an 8 x W-world env, the gather of the packed records
    inside the timed region -- as a dry run on ONE device over gloo (no 8-GPU node is available to the builder: this is the
    closest execution of the N = 8 code path; RCCL itself runs first on the driver's node)."""
    line = _run(["--gpus", "8", "--backend", "gloo", "--share-device", "--steps", "20", "--warmup", "5", "--worlds", "1024", "--reps", "3",
                 "--no-cpu-baseline", "--no-full-loop", "--no-configs3", "--no-pmc", "--no-fresh-scenarios"], 1500)
    assert line["n_gpus"] == 8 and line["steps"] == 20 and line["scaling"] == "weak" and line["value"] > 0
    g = line["extra"]["configs2_gather"]
    assert "error" not in g, g
    assert g["comm_init_per_rank"] == ["ok"] * 8 and g["transport"] == "torch" and "all_gather" in g["path"]
    assert g["bytes_received_per_rank_per_step"] == 7 * 1024 * 4 * 29 * 4 and g["steps_per_launch"] == 20
    assert g["value_is"] == "all" and g["forms"]["root"]["receivers"].startswith("rank 0") and g["forms"]["none"]["agent_steps_per_s"] > 0
    t = line["timing"]
    assert t["timed_reps"] == 3 and t["ms_per_step_min"] <= t["ms_per_step_median"] <= t["ms_per_step_max"]
    assert t["preroll_steps"] == 251 and t["restarts_in_timed_region"] > 0      # the auto-reset path is inside the timed region


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["root", "none"])
def test_bench_two_rank_dry_run_other_gather_forms(mode):
    line = _run(["--gpus", "2", "--backend", "gloo", "--share-device", "--steps", "40", "--warmup", "8", "--worlds", "1024", "--reps", "3",
                 "--gather", mode, "--gather-every", "8", "--no-cpu-baseline", "--no-full-loop", "--no-configs3", "--no-pmc",
                 "--no-fresh-scenarios"], 900)
    g = line["extra"]["configs2_gather"]
    assert "error" not in g, g
    assert g["value_is"] == mode and g["steps_per_launch"] == 8 and g["forms"][mode]["agent_steps_per_s"] == line["value"]
    assert ("rank 0" in line["config"]["parallelism"]) if mode == "root" else ("no data-path collective" in line["config"]["parallelism"])
    assert line["value_is"] == ("root" if mode == "root" else "shard_only")
    assert line["value_shard_only"] == g["forms"]["none"]["agent_steps_per_s"] and line["wire_bound_agent_steps_per_s"] > 0


@pytest.mark.gpu
def test_bench_one_gpu_with_the_exchange_through_forced_rccl():
    """N = 1 development form: the gather inside the timed region through a forced one-rank RCCL communicator (ncclAllGather on the
    communicator's stream) -- the N > 1 code path of the bench with the native transport, on the one GPU there is."""
    line = _run(["--gpus", "1", "--force-rccl", "--steps", "40", "--warmup", "8", "--worlds", "2048", "--reps", "5", "--no-cpu-baseline",
                 "--no-full-loop", "--no-configs3", "--no-pmc", "--no-fresh-scenarios"], 900)
    g = line["extra"]["configs2_gather"]
    assert "error" not in g, g
    assert g["transport"] == "native" and g["uses_rccl"] and g["rccl_version"] >= 20000 and "ncclAllGather" in g["path"]
    assert g["comm_init_per_rank"] == ["ok"] and g["forms"]["all"]["agent_steps_per_s"] == line["value"]
    assert g["forms"]["all"]["bytes_per_link_per_step"] == 0          # one rank: nothing crosses a link


@pytest.mark.gpu
def test_the_json_line_is_the_last_line_of_stdout_even_with_rccl_banners():
    """RCCL prints a version banner through C stdio at its first communicator; with stdout a pipe it is buffered and used to surface at
    process exit, BEHIND the JSON line.  The line must be the last thing a run prints."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--force-rccl", "--steps", "20", "--warmup", "5", "--worlds", "1024", "--reps", "3",
                          "--no-cpu-baseline", "--no-full-loop", "--no-configs3", "--no-pmc", "--no-fresh-scenarios"], env=env, cwd=ROOT, timeout=900,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert lines and lines[-1].startswith("{") and json.loads(lines[-1])["n_gpus"] == 1, lines[-3:]


@pytest.mark.gpu
@pytest.mark.parametrize("mode,says", [("pool", "pre-generated outside the timed region"), ("instep", "generated inside the step kernel")])
def test_bench_other_scenario_sources_as_the_headline(mode, says):
    """`--scenarios pool | instep`: the hashed pool (rounds 1-4's headline) and the in-step generator stay selectable; the line says which"""
    line = _run(["--gpus", "1", "--scenarios", mode, "--steps", "40", "--warmup", "8", "--worlds", "2048", "--reps", "3", "--no-cpu-baseline",
                 "--no-full-loop", "--no-configs3", "--no-pmc", "--no-fresh-scenarios"], 900)
    assert line["config"]["scenarios"] == mode and says in line["config"]["workload"]
    assert line["value"] > 0 and line["timing"]["restarts_in_timed_region"] > 0


@pytest.mark.gpu
def test_bench_falls_back_to_the_pool_when_the_lookahead_cannot_be_had():
    """a look-ahead ring the library refuses (here: more than 4096 steps per launch asked for) must not cost the contract line: the
    headline falls back to the pre-generated pool and says so"""
    line = _run(["--gpus", "1", "--slices", "4100", "--steps", "20", "--warmup", "5", "--worlds", "512", "--reps", "3", "--no-cpu-baseline",
                 "--no-full-loop", "--no-configs3", "--no-pmc", "--no-fresh-scenarios"], 900)
    assert line["config"]["scenarios"] == "pool" and "pre-generated" in line["config"]["workload"]
    assert "failed" in line["extra"]["scenario_fallback"] and line["value"] > 0


@pytest.mark.gpu
def test_the_default_one_gpu_line_carries_its_evidence_and_the_mfma_rooflines():
    """Round 6: the N = 1 line holds `roofline.traffic` (PMC), `cpu_baseline` and -- for BASELINE configs[4] -- MFMA rooflines of the policy
    kernel and of the fused actor kernel that FOLLOW the kernels: issued flop = the matrix instructions the hardware counted
    (SQ_INSTS_MFMA) x 16 384, and the static count bench.py derives from the kernel's loop structure agrees with that count exactly for
    the stand-alone kernel (every row tile computed) and to within the live-row statistics for the actor kernel."""
    line = _run(["--gpus", "1", "--steps", "20", "--warmup", "5", "--reps", "3", "--cpu-seconds", "1", "--no-configs3", "--no-fresh-scenarios"], 900)
    r = line["roofline"]
    assert r["traffic"] and 0.5 < r["traffic_over_moved"] < 2.0 and r["one_step_launch"]["traffic"]
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 1
    ev = line["extra"]["evidence"]
    assert "pmc_traffic" in ev["ran"] and "pmc_mfma" in ev["ran"] and ev["spent_s"] <= 75.0, ev
    fl = line["extra"]["full_ga3c_loop"]
    assert fl["configs4_is"].startswith("actors_only_actor_kernel") and "beyond_configs4_with_trainer" in fl
    pk = fl["policy_kernel"]
    assert pk["mfma_count_source"].startswith("rocprofv3") and pk["mfma_instructions_per_launch"] == pk["mfma_instructions_per_launch_static"] == 2048 * 1144
    assert abs(pk["issued_TFLOPs"] - pk["mfma_instructions_per_launch"] * 16384 / pk["kernel_us"] * 1e-6) < 1e-6 * pk["issued_TFLOPs"]
    assert abs(pk["frac"] - pk["issued_TFLOPs"] / 2500.0) < 1e-9 and 0.15 < pk["frac"] < 0.6
    ar = fl["actors_only_actor_kernel"]["roofline"]
    assert ar["kernel"].startswith("cavoid::actor_kernel<4") and ar["bound"] == "mfma" and 20.0 < ar["kernel_us_per_env_step"] < 60.0
    assert 0.4 < ar["live_row_fraction"] < 0.9 and abs(sum(ar["tiles_by_row_tiles_computed"].values()) - 1.0) < 1e-9
    assert abs(ar["mfma_instructions_per_env_step"] / ar["mfma_instructions_per_env_step_static"] - 1.0) < 0.08, ar
    assert abs(ar["frac"] - ar["mfma_instructions_per_env_step"] * 16384 / ar["kernel_us_per_env_step"] * 1e-6 / 2500.0) < 1e-9
    assert 0.2 < ar["frac"] < 0.6 and ar["useful_f32_grade_TFLOPs"] > 100.0
