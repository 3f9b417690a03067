"""Test infrastructure: run oracle jobs in a child process (tests fan the slow fp64 oracle out over the host cores).
usage: python oracle_worker.py JOBS.pkl OUT.pkl   -- JOBS: list of dicts for conftest.oracle_job; OUT: list of results."""
import os
import pickle
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

if __name__ == "__main__":
    import conftest
    jobs = pickle.load(open(sys.argv[1], "rb"))
    out = [conftest.oracle_job(j) for j in jobs]
    with open(sys.argv[2] + ".tmp", "wb") as f:
        pickle.dump(out, f)
    os.replace(sys.argv[2] + ".tmp", sys.argv[2])
