"""CPU tests of the drop-in boundary: libcaf.so loads, exports exactly what include/caf.h declares,
the Python host imports, and the product path fails loudly (no fallback) without a GPU."""

import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "caf.h")).read()
    return sorted(set(re.findall(r"CAF_EXPORT\s+int32_t\s+(caf_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from pydsproutines_amd import _lib

    lib = _lib.load()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libcaf.so does not export %s" % n
    # the Python binding table and the header agree
    assert sorted(_lib.EXPORTED_SYMBOLS) == names
    assert lib.caf_abi_version() >> 16 == 1


def test_struct_layouts_match_header():
    from pydsproutines_amd import _lib

    # caf_plan_desc: 2 i32, ptr, 2 i32, 2 ptr, 2 i32, ptr, i32(+pad), ptr, i64, 4 i32
    assert ctypes.sizeof(_lib.CafPlanDesc) == 96
    assert _lib.CafPlanDesc.engine.offset == 88
    assert ctypes.sizeof(_lib.CafOutputs) == 7 * ctypes.sizeof(ctypes.c_void_p)  # (never grows: old clients pass exactly this)
    assert ctypes.sizeof(_lib.CafOutputs2) == 11 * ctypes.sizeof(ctypes.c_void_p) and _lib.CafOutputs2.d_surface_t.offset == 56
    assert _lib.CafPlanDesc.max_rx_len.offset == 72


def test_host_modules_import_and_validate_without_gpu():
    from pydsproutines_amd import _lib, xcorrRoutines as X
    from pydsproutines_amd.signalCreationRoutines import makeFreq, randPSKsyms
    from pydsproutines_amd.spectralRoutines import next_fast_len
    from pydsproutines_amd.timingRoutines import Timer
    from pydsproutines_amd.verifyRoutines import compareValues

    np.testing.assert_array_equal(makeFreq(8, 8.0), [0, 1, 2, 3, -4, -3, -2, -1])
    syms, bits = randPSKsyms(16, 4, dtype=np.complex64)
    assert syms.dtype == np.complex64 and np.allclose(np.abs(syms), 1)
    assert next_fast_len(4097) == 4116 and next_fast_len(30) == 30
    raw, frac = compareValues(np.array([1.0, 2.0]), np.array([1.0, 2.5]), verbose=False)
    assert raw == 0.5 and frac == 0.25
    t = Timer()
    t.start()
    assert t.end(showSteps=False) >= 0
    assert X.argmax2d(np.array([[1, 5], [7, 2]])) == (1, 0)
    assert X.convertEffSNRtoQF2(X.convertQF2toEffSNR(0.3)) == pytest.approx(0.3)
    assert X.calcQF2(np.array([1, 1j]), np.array([1, 1j])) == pytest.approx(1.0)
    assert [r[1:] for r in X._runs([3, 4, 5, 9, 12, 15, 20])] == [(3, 1, 3), (9, 3, 3), (20, 1, 1)]
    if _lib.device_count() == 0:
        # the product path must fail loudly when there is no GPU: no CPU fallback exists
        with pytest.raises(RuntimeError):
            X.fastXcorr(np.ones(8, np.complex64), np.ones(64, np.complex64))
        with pytest.raises(RuntimeError):
            X.GroupXcorr(np.ones(32, np.complex64), np.array([0]), np.array([8]), np.array([0.0]), 1.0).xcorr(
                np.ones(64, np.complex64))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "pydsproutines_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(import|from)\s+oracle\b", src, re.M), fn


def test_one_rocm_runtime_per_process_in_either_import_order():
    """libcaf then torch, and torch then libcaf, each leave ONE libamdhip64 / libhsa-runtime64 / librocfft / librccl
    in the process and exit cleanly (two copies abort in free() at interpreter exit and break torch's device probe)."""
    import subprocess
    import sys

    body = (
        "import sys; sys.path.insert(0, %r)\n"
        "%s\n"
        "import collections\n"
        "seen = collections.defaultdict(set)\n"
        "for line in open('/proc/self/maps'):\n"
        "    path = line.split()[-1]\n"
        "    base = path.rsplit('/', 1)[-1]\n"
        "    for stem in ('libamdhip64.so', 'libhsa-runtime64.so', 'librocfft.so', 'librccl.so'):\n"
        "        if base.startswith(stem):\n"
        "            seen[stem].add(path)\n"
        "assert seen['libamdhip64.so'], 'no HIP runtime loaded?'\n"
        "dup = {k: sorted(v) for k, v in seen.items() if len(v) > 1}\n"
        "assert not dup, dup\n"
        "print('single runtime')\n"
    )
    first = "from pydsproutines_amd import _lib; _lib.load(); import torch"
    second = "import torch; from pydsproutines_amd import _lib; _lib.load()"
    for order in (first, second):
        r = subprocess.run([sys.executable, "-c", body % (ROOT, order)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (order, r.stdout, r.stderr)
        assert "single runtime" in r.stdout


def test_header_is_plain_c_and_the_library_links_from_c():
    """include/caf.h compiles as strict C99 and examples/c_client links against libcaf.so with gcc (no C++, no HIP
    headers on the client side): the drop-in boundary is a C-ABI, not a C++ one."""
    import subprocess

    d = os.path.join(ROOT, "examples", "c_client")
    subprocess.run(["make", "-C", d, "clean"], check=True, capture_output=True)
    r = subprocess.run(["make", "-C", d], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "warning" not in (r.stdout + r.stderr).lower()
    assert os.path.exists(os.path.join(d, "caf_client"))


def test_perdelay_jit_planner_and_compile_without_a_gpu():
    """The planner of the run-time-compiled per-delay kernel (caf_jit.hip) runs on the host: every 7-smooth length has a plan whose
    row images fit the LDS and whose simulated bank-conflict cycles stay a small fraction; one length is also compiled for gfx950
    through hiprtc (no GPU needed for that) when hiprtc is installed."""
    import ctypes as ct

    from pydsproutines_amd import _lib

    lib = _lib.load()

    def describe(n, arch=None):
        buf = ct.create_string_buffer(2048)
        rc = lib.caf_perdelay_jit_describe(n, arch, None, buf, 2048)
        return rc, dict(kv.split("=", 1) for kv in buf.value.decode().split(" ") if "=" in kv)

    for n in (96, 360, 1200, 1400, 2401, 5000, 12000, 16200, 4096):
        rc, d = describe(n)
        assert rc == 0 and int(d["n"]) == n, (n, d)
        rad = [int(r) for r in d["radices"].split(",")]
        assert int(np.prod(rad)) == n and 2 <= len(rad) <= 5
        assert int(d["lds_bytes"]) <= 160 * 1024 - 64
        assert int(d["conflict_cycles"]) <= 0.6 * int(d["base_cycles"]), d
    rc, d = describe(97)  # (a prime: Bluestein's chirp transform on a 7-smooth image of at least 2 n - 1 points)
    assert rc == 0 and int(d["bluestein"]) >= 2 * 97 - 1 and int(np.prod([int(r) for r in d["radices"].split(",")])) == int(d["bluestein"])
    rc, d = describe(65536)  # (longer than one LDS image: four residues of a 16384-point transform per row)
    assert rc == 0 and int(d["q"]) == 4 and int(np.prod([int(r) for r in d["radices"].split(",")])) == 16384
    rc, d = describe(10007)  # (a prime whose chirp transform does not fit the LDS: no plan, the rows path keeps it)
    assert rc == 0 and not d
    rc, d = describe(1200, b"gfx950")
    if rc == 0:
        assert int(d["code_bytes"]) > 4000
    else:  # no hiprtc on this box: the library says so and the prebuilt kernels stay in charge
        assert "hiprtc" in _lib.last_error()
