"""Register / scratch / LDS use of every kernel and out-of-line device function of libcaf, from the compiler's own assembly
listing (hipcc -S --cuda-device-only of each translation unit).  Spills = "Folded Spill" stores in the listing; scratch that
is not spills is the callee-saved register block of a noinline role function (saved once per call, i.e. once per work item).
usage: python scripts/kernel_resources.py [file.hip ...]   (default: every .hip of pydsproutines_amd/csrc)"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pydsproutines_amd", "csrc")


def listing(src):
    with tempfile.NamedTemporaryFile(suffix=".s", delete=False) as f:
        out = f.name
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                    "-I" + CSRC, "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
    os.unlink(out)
    return text


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines()


def functions(text):
    """(name, vgprs, sgprs, scratch bytes, spill stores, spill reloads, in-loop spill ops, lds bytes) per function."""
    rows = []
    lines = text.splitlines()
    starts = [i for i, ln in enumerate(lines) if re.match(r"^_Z\w+:", ln)]
    for k, i in enumerate(starts):
        name = lines[i].split(":")[0]
        end = starts[k + 1] if k + 1 < len(starts) else len(lines)
        body = lines[i:end]
        info = {}
        for ln in body:
            m = re.match(r";\s*(NumVgprs|NumSgprs|ScratchSize|LDSByteSize):\s*(\d+)", ln.strip())
            if m and m.group(1) not in info:
                info[m.group(1)] = int(m.group(2))
        if "NumVgprs" not in info:
            continue
        spill = sum("Folded Spill" in ln for ln in body)
        reload_ = sum("Folded Reload" in ln for ln in body)
        # spill traffic inside loops: between a loop header comment and the function end, lines tagged "in Loop"
        depth_tagged = 0
        inloop = False
        for ln in body:
            if "Loop Header" in ln or "in Loop:" in ln:
                inloop = True
            if inloop and ("Folded Spill" in ln or "Folded Reload" in ln):
                depth_tagged += 1
            if re.match(r"^\.LBB\d+_\d+:\s*$", ln) and "Loop" not in ln:
                inloop = False
        rows.append((name, info.get("NumVgprs", 0), info.get("NumSgprs", 0), info.get("ScratchSize", 0), spill, reload_, depth_tagged,
                     info.get("LDSByteSize", 0)))
    return rows


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    print("%-96s %5s %5s %8s %7s %8s %9s %8s" % ("function", "VGPR", "SGPR", "scratch", "spills", "reloads", "in loops", "LDS"))
    for f in files:
        rows = functions(listing(f))
        names = demangle([r[0] for r in rows])
        for r, n in zip(rows, names):
            n = n.replace("caf::", "").replace("(anonymous namespace)::", "")
            n = re.sub(r"\(.*", "", n)
            if r[3] or r[4] or "k_" in n or "persistent_" in n:
                print("%-96s %5d %5d %8d %7d %8d %9d %8d" % (n[:96], r[1], r[2], r[3], r[4], r[5], r[6], r[7]))


if __name__ == "__main__":
    main()
