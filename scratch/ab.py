"""A/B of two builds of the library on the decode bench: python scratch/ab.py <lib> <workload>."""
import os, sys, json, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tansu_amd import hip
hip.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py", "--workload", sys.argv[2], "--no-cpu-baseline", "--steps", "30"]
import runpy
runpy.run_path(os.path.join(os.path.dirname(__file__), "..", "bench.py"), run_name="__main__")
