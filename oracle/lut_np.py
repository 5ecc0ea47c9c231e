"""ORACLE (test infrastructure only) -- numpy restatement of `resample_kernel` (amico/lut.pyx:274-311).

Only tests/ and bench.py may import this.  The reference statement is numpy itself (a float32 `np.dot` per LUT
orientation writing into an array of ones), so this is the reference's arithmetic up to the BLAS summation order.
"""
import numpy as np


def resample_kernel(KRlm, nS, idx_out, Ylm_out, is_isotropic, ndirs):
    if not is_isotropic:
        KR = np.ones((ndirs, nS), dtype=np.float32)
        for i in range(ndirs):
            KR[i, idx_out] = np.dot(Ylm_out, KRlm[i, :]).astype(np.float32)
    else:
        KR = np.ones(nS, dtype=np.float32)
        KR[idx_out] = np.dot(Ylm_out, KRlm).astype(np.float32)
    return KR
