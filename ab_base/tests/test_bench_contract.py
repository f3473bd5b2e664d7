"""bench.py's measurement arithmetic (no GPU): the algorithmic bytes behind ``roofline.achieved`` are the
per-image figures of SURVEY.md section 8d times the images of a launch, and the command-line contract
(``--gpus / --steps / --warmup``, defaults that finish in minutes) is what the driver calls."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_bytes_are_the_survey_figures(bench):
    NV = 100 * 167 + 50 * 84 + 25 * 42 + 13 * 21
    assert bench.NV == NV == 22223
    mb = lambda *a: bench.msda_algorithmic_bytes(*a) / 1e6                     # noqa: E731
    # encoder fwd 79.65 MB/img, decoder fwd 23.52, encoder bwd 136.6 (136.5 exactly), decoder bwd 46.7
    assert mb("fwd", 1, NV, 4) == pytest.approx(79.65, abs=0.01)
    assert mb("fwd", 1, 300, 4) == pytest.approx(23.52, abs=0.01)
    assert mb("bwd", 1, NV, 4) == pytest.approx(136.54, abs=0.01)
    assert mb("bwd", 1, 300, 4) == pytest.approx(46.74, abs=0.01)
    # bf16 value / out / grad_out, fp32 loc / attn / grad_value: encoder fwd 56.9 MB/img; B images scale linearly
    assert mb("fwd", 1, NV, 2) == pytest.approx(56.89, abs=0.01)
    assert mb("bwd", 4, NV, 2) == pytest.approx(4 * mb("bwd", 1, NV, 2))
    # by definition: value + loc (1024 B) + attn (512 B) + out per query; bwd adds grad_value (fp32) + grad_loc + grad_attn
    assert bench.msda_algorithmic_bytes("fwd", 2, 300, 2) == 2 * (NV * 512 + 300 * (1024 + 512 + 512))
    assert bench.msda_algorithmic_bytes("bwd", 2, 300, 2) - bench.msda_algorithmic_bytes("fwd", 2, 300, 2) == \
        2 * (NV * 1024 + 300 * 1536)
    # fused prologue: the projection output (384 values) + 4 reference points replace loc / attn
    assert bench.msda_algorithmic_bytes("fwd_fused", 1, NV, 2) == NV * 512 + NV * (384 * 2 + 32) + NV * 512
    assert bench.HBM_PEAK_GBS == 8000.0


def test_command_line_contract(bench, monkeypatch, capsys):
    import sys
    monkeypatch.setattr(sys, "argv", ["bench.py", "--help"])
    with pytest.raises(SystemExit):
        bench.main()
    text = capsys.readouterr().out
    for flag in ("--gpus", "--steps", "--warmup", "--batch", "--dtype"):
        assert flag in text
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'add_argument("--gpus", type=int, default=1)' in src          # no flags: one GPU, a K / W that finish in minutes
    assert 'add_argument("--steps", type=int, default=20)' in src and 'add_argument("--warmup", type=int, default=3)' in src
    assert '"higher_is_better": True' in src and '"scaling": "weak"' in src and '"vs_baseline": None' in src


def _run_bench(args, env_extra=None, timeout=240):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_gpus_2_without_a_launcher_starts_two_ranks():
    """``python bench.py --gpus 2`` with no WORLD_SIZE in the environment must start its ranks itself
    (the driver's command shape; the reference: tools/dist_train_increment.sh:22-28).  ``--launch-check`` runs only the
    rank plumbing -- on this CPU-only host over gloo -- and prints the all-reduced rank count."""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                  # rank 0 alone prints
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out.get("gloo_ranks", out.get("rccl_ranks")) == 2


def test_a_failing_rank_fails_the_launch():
    """The parent must not report success (or hang) when a rank dies: a WORLD_SIZE that disagrees with --gpus makes
    every child exit 1 at once; so does a real run on a host without GPUs."""
    r = _run_bench(["--gpus", "2", "--launch-check"], {"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr
    import torch
    if not torch.cuda.is_available():
        r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"DSKD_BENCH_REHEARSE": "1"})
        assert r.returncode != 0 and "needs a GPU" in r.stderr


def test_launcher_form_still_works():
    """``python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`` (the documented form) keeps working."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29731", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--launch-check"], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    assert '"n_gpus": 2' in r.stdout
