"""Dump the structured datasets of one HDF5 file to .npy files (golden generation only).

TEST INFRASTRUCTURE.  Run with an interpreter that has h5py (in the build
container: /opt/conda/bin/python3.9); called by oracle/make_golden.py.

usage: _h5dump.py in.h5 dataset out.npy
"""
import sys

import h5py
import numpy as np

fn, dset, out = sys.argv[1:4]
with h5py.File(fn, 'r') as f:
    np.save(out, f[dset][:])
