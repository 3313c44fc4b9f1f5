#!/usr/bin/env python
"""WavEncoder timing at the S2G size (out_dim=1536, audio_in=2, 196 frames ~ 105 840 samples)."""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
from motioncraft_amd.wav_encoder import NativeWavEncoder
from motioncraft_amd.synthetic import make_wav_encoder_state

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 105840
D, CIN = 1536, 2
enc = NativeWavEncoder(D, CIN, make_wav_encoder_state(D, CIN, 0))
wav = torch.randn(B, S, CIN, device='cuda')
spec = [(CIN, D // 4, 5, 1600, 1), (D // 4, D // 4, 6, 0, 1), (D // 4, D // 4, 1, 7, 0), (D // 4, D // 2, 6, 0, 1),
        (D // 2, D // 2, 1, 7, 0), (D // 2, D, 3, 0, 1)]
T, fl = S, 0
for cin, pl, st, pad, down in spec:
    T = (T + 2 * pad - 15) // st + 1
    fl += T * pl * 15 * 2 * (cin * (1 + down) + pl)
out = enc(wav)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    out = enc(wav)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print(f'WavEncoder B={B} samples={S}: frames {out.shape[1]}, {fl/1e9:.1f} GFLOP/sample, {ms:.2f} ms -> {B*fl/ms/1e9:.1f} TFLOP/s '
      f'({ms/B:.2f} ms/sample)')
