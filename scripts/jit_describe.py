"""Plan, LDS layout and (optionally) a compile of the run-time per-delay kernel for given cutout lengths -- no GPU needed.
usage: python scripts/jit_describe.py [--compile] [--dump DIR] N [N ...]"""
import ctypes as ct
import os
import sys

sys.path.insert(0, ".")
from pydsproutines_amd import _lib  # noqa: E402

args = sys.argv[1:]
do_compile = "--compile" in args
dump = None
if "--dump" in args:
    dump = args[args.index("--dump") + 1]
    args.remove("--dump"), args.remove(dump)
    os.makedirs(dump, exist_ok=True)
    do_compile = True
args = [a for a in args if a != "--compile"]
lib = _lib.load()
for n in (int(a) for a in args):
    buf = ct.create_string_buffer(2048)
    path = os.path.join(dump, "pdj_%d.hsaco" % n).encode() if dump else None
    rc = lib.caf_perdelay_jit_describe(n, b"gfx950" if do_compile else None, path, buf, 2048)
    print(buf.value.decode() if rc == 0 else "n=%d: rc=%d %s" % (n, rc, _lib.last_error()))
