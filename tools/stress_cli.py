"""Repeat the CLI round trip of the smpte golden in fresh processes and count mismatches (race hunting)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc
ENC, DEC = os.path.join(ROOT, "bin", "encode"), os.path.join(ROOT, "bin", "decode")
src = open(os.path.join(orc.GOLDEN, "smpte.pnm"), "rb").read()
want = open(os.path.join(orc.GOLDEN, "smpte.dwt"), "rb").read()
pix = orc.read_pnm(os.path.join(orc.GOLDEN, "smpte.pnm")).tobytes()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
bad_e = bad_d = 0
for i in range(N):
    r = subprocess.run([ENC, "-", "-"], input=src, capture_output=True, timeout=120)
    if r.returncode or r.stdout != want:
        bad_e += 1
        d = next((k for k in range(min(len(want), len(r.stdout))) if want[k] != r.stdout[k]), -1)
        print(f"iter {i}: encode rc={r.returncode} len={len(r.stdout)} want={len(want)} first diff at byte {d}", flush=True)
    r2 = subprocess.run([DEC, "-", "-"], input=want, capture_output=True, timeout=120)
    body = r2.stdout[len(b"P6 320 240 255\n"):]
    if r2.returncode or body != pix:
        bad_d += 1
        nd = sum(1 for a, b in zip(body, pix) if a != b)
        print(f"iter {i}: decode rc={r2.returncode} len={len(body)} differing bytes={nd} stderr={r2.stderr[-200:]!r}", flush=True)
print(f"{N} iterations: encode mismatches {bad_e}, decode mismatches {bad_d}")
