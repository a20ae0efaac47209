"""Run another script of this repo against another build of the library:  python scripts/exp/with_lib.py LIB SCRIPT [args]"""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import deeploopcloser_amd._lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
