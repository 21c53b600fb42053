"""Static check of the gfx950 ISA hipcc emits for our kernels: a VMEM store of more than 64 bits whose data VGPRs are overwritten by a
VALU instruction within the next `WIN` instructions. LLVM's hazard recognizer only guards this write-after-read hazard when the store
has NO SGPR soffset (GCNHazardRecognizer::createsVALUHazard); with an SGPR soffset it assumes the hardware is safe -- on MI355X under
load it is not (k_ig3r, tools/diag_ig3r.py: lanes 12-15 of the second data dword came out overwritten). Usage:
    hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S x.hip -o x.s ; python tools/scan_store_hazard.py x.s [...]"""
import re
import sys

WIN = 2
STORE = re.compile(r"^\s+(buffer_store_dwordx[34]|global_store_dwordx[34]|flat_store_dwordx[34]|scratch_store_dwordx[34])\s+(.*)$")
VREG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def regs(tok):
    m = VREG.fullmatch(tok.strip())
    if not m:
        return set()
    if m.group(1) is not None:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return {int(m.group(3))}


def dst_regs(line):
    m = re.match(r"^\s+(v_[a-z0-9_]+)\s+([^,]+)(?:,\s*([^,]+))?", line)
    if not m:
        return set()
    op = m.group(1)
    if op.startswith("v_cmp") or op.startswith("v_mfma") or op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
        return set()
    d = regs(m.group(2))
    if op.startswith("v_permlane16_swap") or op.startswith("v_permlane32_swap") or op.startswith("v_swap"):
        d |= regs(m.group(3) or "")
    return d


def is_inst(line):
    return line.startswith("\t") and not line.startswith("\t.") and not line.startswith("\t;")


def scan(path):
    kernel, lines = None, open(path).read().split("\n")
    found = 0
    for i, line in enumerate(lines):
        km = re.match(r"^(_Z\w+):", line)
        if km:
            kernel = km.group(1)
        m = STORE.match(line)
        if not m:
            continue
        ops = [o.strip() for o in m.group(2).split(",")]
        # buffer_store: vdata first; global_store: vaddr, vdata
        data = regs(ops[0]) if m.group(1).startswith("buffer") or m.group(1).startswith("scratch") else regs(ops[1])
        k, j = 0, i + 1
        while k < WIN and j < len(lines):
            if is_inst(lines[j]):
                k += 1
                if lines[j].lstrip().startswith("s_nop") or lines[j].lstrip().startswith("s_waitcnt"):
                    break                      # (an s_nop of any length is what we insert by hand as the guard)
                hit = dst_regs(lines[j]) & data
                if hit:
                    found += 1
                    print(f"{path}:{i + 1}: {kernel}: `{line.strip()}` data v{sorted(hit)} overwritten {k} instruction(s) later by `{lines[j].strip()}`")
                    break
            j += 1
    return found


if __name__ == "__main__":
    total = sum(scan(p) for p in sys.argv[1:])
    print(f"{total} hazard site(s)")
    sys.exit(1 if total else 0)
