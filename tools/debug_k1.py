import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_amd import synth
A, B, _ = synth.make_pair(300, 300, 150, seed=77)
B[10] = B[20]; B[200:210] = A[5]; A[100] = A[101]
A.tofile("/tmp/A.bin"); B.tofile("/tmp/B.bin")
exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "k1_debug")
for chunk in (256, 512):
    print("chunk", chunk, flush=True)
    subprocess.run([exe, "/tmp/A.bin", "300", "/tmp/B.bin", "300", str(chunk)], check=False)
A, B, _ = synth.make_pair(2048, 2048, 700, seed=3)
A.tofile("/tmp/A.bin"); B.tofile("/tmp/B.bin")
for chunk in (256, 2048):
    print("2048 chunk", chunk, flush=True)
    subprocess.run([exe, "/tmp/A.bin", "2048", "/tmp/B.bin", "2048", str(chunk)], check=False)
