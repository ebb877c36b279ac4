"""bench.py / train_ppo.py host logic that needs no GPU: the --gpus spawner, the world-size check, the roofline
arithmetic and the reference's README command line."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(kw)
    return e


def test_algorithmic_bytes_formula():
    sys.path.insert(0, ROOT)
    import bench
    # Solo12 walk, h = 1 -> 4*(12 + 74 + 96 + 76 + 90 + 2) + 1 = 1401 B
    assert bench.algorithmic_bytes_per_env_step(A=12, S=37, C=48, O=76, hist_state=38) == 1401
    assert bench.VALU_ISSUE_PEAK == pytest.approx(1228.8e9)       # 256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles


def test_gpus_flag_spawns_ranks_and_propagates_failure():
    """`python bench.py --gpus 2` started directly must launch two ranks itself (torch.distributed.run children,
    before anything touches a GPU).  Here there is no GPU: both ranks die with the engine's "needs a HIP device",
    the parent reports the failed 2-rank run and exits non-zero without printing a result line."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0",
                        "--ppo-steps", "0", "--no-cpu-baseline"], env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert "2-rank run failed" in p.stderr
    assert '"metric"' not in p.stdout
    assert "needs a HIP device" in (p.stdout + p.stderr)          # the children really started and got as far as the device check


def test_gpus_flag_must_match_world_size():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in p.stderr


def test_reference_readme_command_line_parses():
    """/root/reference/README.md:34 verbatim (paths aside): every flag of training/train_ppo.py:9-45 is accepted."""
    sys.path.insert(0, ROOT)
    import train_ppo
    a = train_ppo.get_ppo_args("--num-agents 64 --logdir /tmp/L --log-interval 5 --save-interval 50 --lr 0.00025 --entropy-coef 0.01 "
                               "--clip-param 0.1 --ppo-epoch 5 --mini-batch-size 512 --clip-value-loss --config-file ./configs/basic.yaml "
                               "--use-gae --use-linear-lr-decay --seed 1".split())
    assert a.num_agents == 64 and a.clip_value_loss and a.use_gae and a.ppo_epoch == 5 and a.mini_batch_size == 512
    b = train_ppo.get_ppo_args("--output-size 64 --timestamp run1 --curriculum-schedule 10 --base-checkpoint x.pt --task walk".split())
    assert b.output_size == 64 and b.timestamp == "run1" and b.curriculum_schedule == 10


@pytest.mark.parametrize("strict,code", [(False, 0), (True, 3)])
def test_watchdog_prints_a_valid_headline_when_an_auxiliary_leg_blocks(strict, code):
    """a PPO leg that never returns (a stuck collective on a multi-GPU node) must not cost the headline.  Simulated with
    a sleeping leg: the guard's timer prints the one JSON line -- headline fields intact, the unfinished leg carrying an error entry,
    `aux_legs_complete` false -- and the process leaves; exit code 0, or 3 with --strict (let the caller decide)."""
    prog = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "g = bench.LineGuard(0, 0.5, %r, lambda st: {'metric': 'env-steps/s', 'value': 1.0, 'ppo_loop': st['ppo'], 'f64': st['f64']}, True)\n"
            "g.start(); g.state['f64'] = {'value': 2.0}\n"
            "time.sleep(60)\n"                      # the PPO leg, hanging
            "print('never reached')\n" % (ROOT, strict))
    p = subprocess.run([sys.executable, "-c", prog], env=_env(), capture_output=True, text=True, timeout=120)
    assert p.returncode == code
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    import json
    line = json.loads(lines[0])
    assert line["metric"] == "env-steps/s" and line["value"] == 1.0 and line["f64"] == {"value": 2.0}
    assert line["aux_legs_complete"] is False and "watchdog" in line["note"] and "did not finish" in line["ppo_loop"]["error"]


def test_line_guard_prints_once_and_only_on_rank_0(capsys):
    sys.path.insert(0, ROOT)
    import bench
    exits = []
    g = bench.LineGuard(0, 3600, False, lambda st: {"metric": "m", "ppo_loop": st["ppo"]}, False, exit_fn=exits.append)
    g.start(); g.finish(); g.emit(); g.on_timeout()
    out = [ln for ln in capsys.readouterr().out.splitlines() if ln.strip()]
    assert len(out) == 1 and '"aux_legs_complete": true' in out[0] and exits == [0]
    g1 = bench.LineGuard(1, 3600, True, lambda st: {"metric": "m"}, True, exit_fn=exits.append)
    g1.finish(); g1.on_timeout()
    assert capsys.readouterr().out.strip() == "" and exits == [0, 3]


def test_rccl_filters_are_exported_before_the_communicator_exists(monkeypatch):
    """SOLORL_RCCL_ALGO / SOLORL_RCCL_PROTO -> NCCL_ALGO / NCCL_PROTO (tree / direct suits the 80 KB bucket); unset leaves
    RCCL's tuner alone, and rccl_env() is what bench.py records."""
    from solorl_amd.ppo import dist as D
    for k in ("NCCL_ALGO", "NCCL_PROTO", "SOLORL_RCCL_ALGO", "SOLORL_RCCL_PROTO"):
        monkeypatch.delenv(k, raising=False)
    assert D.pin_rccl_from_env() == {}
    monkeypatch.setenv("SOLORL_RCCL_ALGO", "Tree"); monkeypatch.setenv("SOLORL_RCCL_PROTO", "LL")
    assert D.pin_rccl_from_env() == {"NCCL_ALGO": "Tree", "NCCL_PROTO": "LL"}
    assert os.environ["NCCL_ALGO"] == "Tree"
    monkeypatch.delenv("NCCL_ALGO"); monkeypatch.delenv("NCCL_PROTO")


def test_line_guard_rearm_shortens_the_fuse():
    """The captured-all-reduce probe (the last thing a multi-GPU bench does) gets its own 60 s fuse: rearm() replaces the running timer."""
    sys.path.insert(0, ROOT)
    import bench
    import time
    fired = []
    g = bench.LineGuard(1, 3600, False, lambda st: {"metric": "m"}, False, exit_fn=fired.append)
    g.start(); g.rearm(0.2)
    time.sleep(0.6)
    assert fired == [0] and g.timeout_s == 0.2
