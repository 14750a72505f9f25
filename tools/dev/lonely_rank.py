import sys, time, os, faulthandler
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
faulthandler.dump_traceback_later(25, exit=False)
import quaff_amd as Q
from quaff_amd.api import QuaffHipError
ctx = Q.Context(0)
ctx.set_params_json(None)
uid = Q.Context.comm_unique_id()
print("got id", flush=True)
t0 = time.time()
try:
    ctx.comm_init_rank(uid, 0, 2)
except QuaffHipError as e:
    print("STATUS", e.code, "%.1f" % (time.time() - t0), str(e), flush=True)
    sys.exit(3)
print("JOINED", flush=True)
