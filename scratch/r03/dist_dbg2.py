import os, sys, faulthandler, socket, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if "RANK" not in os.environ:
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ps = [subprocess.Popen([sys.executable, "-u", __file__], env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))) for r in range(2)]
    for p in ps:
        try: p.wait(timeout=150)
        except subprocess.TimeoutExpired: p.kill()
    sys.exit(0)
faulthandler.dump_traceback_later(60, exit=True)
import test_dist_gpu as T
class Q:
    def put(self, x): print("RESULT", x[0], [r[:2] for r in x[1]], flush=True)
T._worker(int(os.environ["RANK"]), 2, int(os.environ["MASTER_PORT"]), Q(), False)
