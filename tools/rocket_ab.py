import sys, os, glob; sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(),'tools'))
import oc_trace
print("--- product (coarse phase in the wide kernel)")
oc_trace.run("rocket", 100, 1024, "f32", 0)
print("--- LFSD_COARSE_MIN_GRID=100000 (no coarse phase)")
oc_trace.run("rocket", 100, 1024, "f32", 0, library=glob.glob("learning-from-sparse-demonstrations_amd/csrc/build/ab_*_nowidecoarse.so")[0])
