"""`python bench.py --gpus N` WITHOUT a launcher must start its N ranks itself (the driver's BENCH command is `python3 bench.py --gpus 1 ...`;
if SCALE is called the same way, the first multi-GPU record must not be lost to "must be launched with torch.distributed.run").  CPU only: the launcher is
exercised with stub workers -- it never imports torch and never touches a device."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(tmp_path, body):
    w = tmp_path / "worker.py"
    w.write_text(textwrap.dedent(body))
    return [sys.executable, str(w)]


def _launch(tmp_path, n, body, **kw):
    code = f"import sys; sys.path.insert(0, {ROOT!r}); import bench; sys.exit(bench.launch_ranks({n}, {_worker(tmp_path, body)!r}, **{kw!r}))"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env, stdin=subprocess.DEVNULL)


def test_the_launcher_gives_every_rank_the_environment_of_torch_distributed_run_and_relays_rank_0(tmp_path):
    r = _launch(tmp_path, 4, """
        import json, os, sys
        e = {k: os.environ[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
        print(json.dumps(e), flush=True)      # rank 0: the line; the others: must land on stderr
        assert "torch" not in sys.modules
    """)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                      # ONE line on stdout: rank 0's
    e = json.loads(lines[0])
    assert e["RANK"] == "0" and e["LOCAL_RANK"] == "0" and e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" and int(e["MASTER_PORT"]) > 0
    others = [json.loads(l) for l in r.stderr.splitlines() if l.startswith("{")]
    assert sorted(o["RANK"] for o in others) == ["1", "2", "3"] and {o["MASTER_PORT"] for o in others} == {e["MASTER_PORT"]}


def test_a_failing_rank_fails_the_launch_and_does_not_leave_the_others_running(tmp_path):
    r = _launch(tmp_path, 3, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(600)                        # a rank waiting for the dead one in a barrier
    """, grace_s=1.0)
    assert r.returncode == 7, (r.returncode, r.stderr)
    assert "rank 1 exited with code 7" in r.stderr


def test_bench_py_becomes_the_launcher_only_without_world_size():
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    head = main[:main.index("import torch")]
    assert '"WORLD_SIZE" not in os.environ and args.gpus > 1' in head and "launch_ranks(" in head   # decided before torch is imported
    launcher = src[src.index("def launch_ranks("):src.index("def main():")]
    assert "import torch" not in launcher and "os.exec" not in launcher and "execv" not in launcher
