"""CPU oracle for the CAF / matched-filter hot path.

TEST INFRASTRUCTURE ONLY.  This package is a NumPy restatement of the
reference's CPU algorithms (icyveins7/pydsproutines: xcorrRoutines.py,
spectralRoutines.py, the custom_kernels/*.cu semantics).  It exists to CHECK
the HIP product path; nothing under ``pydsproutines_amd/`` may import it.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg are allowed to use it.

Parity pinning: every function here is checked against the importable Python
reference in the build container (``tests/golden/make_golden.py``) and against
the committed golden vectors / known-answer tests in ``tests/golden/``
(SURVEY.md Appendix A).  The native C++/IPP twins of the reference
(CyIppXcorrFFT, CyGroupXcorrFFT, pbIppCZT32fc, pbIppGroupXcorrCZT) are NOT
buildable here (Intel IPP and the un-vendored ``ipp_ext`` submodule are
absent), so those are restated from source text and pinned to the Python
oracle, which computes the same mathematics.  ``oracle/c/ippxcorrfft_port.c``
(bound by ``oracle/cport.py``) is such a restatement in plain C with pthreads
-- the threaded CPU baseline of bench.py -- pinned by
``tests/test_oracle_c_port.py`` to KAT-2 and to the NumPy oracle.
"""

from .spectral import makeFreq, next_fast_len, czt, CZTCached, dft  # noqa: F401
from .xcorr import (  # noqa: F401
    fastXcorr,
    cztXcorr,
    caf_bins,
    caf_overlap_save,
    GroupXcorr,
    GroupXcorrFFT,
    GroupXcorrCZT,
    GroupXcorrCZT_Permutations,
    GroupXcorrGPU,
    GenXcorr,
    fineFreqTimeSearch,
    makeTimeScanSteervec,
    TemplateCrossCorrelator,
    IppXcorrFFT,
    IppGroupXcorrFFT,
    cp_fastXcorr,
    cp_fastXcorr_v2,
    argmax2d,
    calcQF2,
)
from . import kernels  # noqa: F401
