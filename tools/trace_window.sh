python - <<'PY'
import gzip, os, sys, subprocess
sys.path.insert(0, ".")
from tests import world_util as wu
open("/tmp/world.bin","wb").write(gzip.open("tests/golden/matcher_world.bin.gz").read())
exe = wu.build_adapter_world("orbx")
for sc in ("proj_mp_mono_th1", "proj_last_mono", "fuse_mono", "bow_kf_frame"):
    r = subprocess.run([exe, "/tmp/world.bin", "/tmp/o.txt", sc, "--time", "/tmp/t.json"], capture_output=True, text=True, env=dict(os.environ, ORBX_TRACE_WINDOW="1"))
    lines = [l for l in r.stderr.splitlines() if "orbx window" in l]
    print(sc, open("/tmp/t.json").read().replace("\n"," "))
    for l in lines[-3:]: print("   ", l)
PY
