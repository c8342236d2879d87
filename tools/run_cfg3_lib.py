"""tools/run_cfg3.py against another build of the library (A/B experiments): run_cfg3_lib.py <lib.so> [args]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
pkg.hbmpc.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [sys.argv[0]] + sys.argv[2:]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "run_cfg3.py")).read())
