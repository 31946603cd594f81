"""developer: run one scenario of tests/test_gpu_graphed_step.py's two-rank child with a stack dump after 90 s (where does it hang?)
   python tools/dev/two_rank_debug.py small_first"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("t", os.path.join(ROOT, "tests", "test_gpu_graphed_step.py"))
src = open(os.path.join(ROOT, "tests", "test_gpu_graphed_step.py")).read()
start = src.index('CHILD2 = r"""') + len('CHILD2 = r"""')
child = src[start:src.index('"""', start)]
code = "import faulthandler, sys\nfaulthandler.dump_traceback_later(90, exit=True, file=sys.stderr)\n" + child % {"root": ROOT, "port": "29671"}
env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
procs = [subprocess.Popen([sys.executable, "-c", code, str(r), sys.argv[1]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(2)]
for r, p in enumerate(procs):
    try:
        o, e = p.communicate(timeout=150)
    except subprocess.TimeoutExpired:
        p.kill(); o, e = p.communicate()
    print(f"==== rank {r} rc {p.returncode}\n{o[-1500:]}\n---- stderr\n{e[-5000:]}")
