"""CPU: static check of the gfx950 ISA hipcc emits for every kernel of the library.

A VMEM store of more than 64 bits whose data registers are overwritten by a VALU instruction right after it is a write-after-read
hazard that hipcc only guards when the store has no SGPR offset. On MI355X the unguarded form corrupted the last tile of every
k_ig3r workgroup at full size (round 2; tools/diag_ig3r.py, nndetection_amd/csrc/conv_igemm.hip: k_ig3r, last-tile epilogue) while all
small-size parity tests passed, so the pattern is checked on the emitted code itself."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_unguarded_store_data_overwrite(tmp_path):
    asm = str(tmp_path / "isa")
    subprocess.check_call(["bash", os.path.join(ROOT, "nndetection_amd", "csrc", "build.sh"), "--asm", asm])
    files = sorted(os.path.join(asm, f) for f in os.listdir(asm) if f.endswith(".s"))
    assert len(files) >= 15, files
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scan_store_hazard.py")] + files, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-4000:]
    assert "0 hazard site(s)" in r.stdout
