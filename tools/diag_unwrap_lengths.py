"""Container-only: deviation of oracle.audio.stft_to_phase_magn from the REFERENCE's (imported from /root/reference) as a
function of track length, with the unwrap's running sum in float32 (rounds 1-2) and in float64 rounded per frame (round 3 ==
torch.cumsum on the CPU).  Prints the table quoted in DESIGN.md section 2.  `python tools/diag_unwrap_lengths.py`"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ref_loader import load_audio  # noqa: E402
from oracle import audio as OA  # noqa: E402


def unwrap_fp32(phi):
    phi = np.asarray(phi, dtype=np.float32)
    dphi = np.zeros_like(phi)
    dphi[:, 1:] = phi[:, 1:] - phi[:, :-1]
    pi, two_pi = np.float32(np.pi), np.float32(2 * np.pi)
    dm = np.mod(dphi + pi, two_pi) - pi
    dm[(dm == -pi) & (dphi > 0)] = pi
    adj = dm - dphi
    adj[np.abs(dphi) < pi] = 0
    return phi + np.cumsum(adj, axis=1, dtype=np.float32)


if __name__ == "__main__":
    ref = load_audio()
    rng = np.random.default_rng(7)
    new_unwrap = OA.unwrap
    print(f"{'frames':>8} {'magn new':>10} {'phase fp32-sum':>15} {'phase fp64-sum':>15} {'max|unwrapped|':>15}")
    for frames in (553, 2001, 20001, 103360):
        wav = rng.random(256 * (frames - 1), dtype=np.float32) - 0.5
        c = OA.stft(wav)
        assert c.shape == (512, frames)
        m_ref, p_ref = ref.stft_to_phase_magn(torch.from_numpy(c))
        OA.unwrap = new_unwrap
        m_new, p_new = OA.stft_to_phase_magn(c)
        OA.unwrap = unwrap_fp32
        _, p_old = OA.stft_to_phase_magn(c)
        OA.unwrap = new_unwrap
        u = np.abs(new_unwrap(np.angle(c).astype(np.float32))).max()
        print(f"{frames:8d} {np.abs(m_new - m_ref.numpy()).max():10.2e} {np.abs(p_old - p_ref.numpy()).max():15.2e} "
              f"{np.abs(p_new - p_ref.numpy()).max():15.2e} {u:15.1f}", flush=True)
