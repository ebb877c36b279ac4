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
