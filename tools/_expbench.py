import os, sys
sys.path.insert(0, os.getcwd())
from ucd_amd import hip
v = os.environ.get("UCD_EXP")
if v: hip.LIB_PATH = os.path.join(os.path.dirname(hip.LIB_PATH), "exp", f"libucd_hip_{v}.so")
sys.argv = ["tools/pixcon_bench.py", "f16", "dom"]
exec(open("tools/pixcon_bench.py").read())
