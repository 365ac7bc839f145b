"""fp32 rounding error of Winograd F(4x4,3x3) beside F(2x2,3x3) and the direct form, against float64, on the layer shapes of the
dominant launches (48->64 @128x128, 64->80 @64x64).  CPU only: every transform and the channel contraction run in float32 (torch CPU
matmul / einsum accumulate in fp32 like the MFMA does), the reference is F.conv2d in float64 on the same float32 inputs.
Prints rms and max error relative to the rms of the exact output.  ("measure F(4x4,3x3) in fp32".)"""
import sys
import torch
import torch.nn.functional as F

torch.manual_seed(0)
torch.set_num_threads(8)

BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)

BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
                    [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
                   [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)


def wino(x, w, BT, G, AT, dt):
    """x (N,C,H,W), w (O,C,3,3) float32 values; all arithmetic in dtype dt.  Tile size m = AT.shape[0], alpha = BT.shape[0]."""
    m, a = AT.shape[0], BT.shape[0]
    n, c, h, wd = x.shape
    xp = F.pad(x.to(dt), (1, 1 + (-wd) % m, 1, 1 + (-h) % m))
    tiles = xp.unfold(2, a, m).unfold(3, a, m)                    # N,C,th,tw,a,a
    BTd, Gd, ATd = BT.to(dt), G.to(dt), AT.to(dt)
    U = Gd @ w.to(dt) @ Gd.T                                       # O,C,a,a   (the product packs this once per update, in fp32 too)
    V = BTd @ tiles @ BTd.T                                        # N,C,th,tw,a,a
    M = torch.einsum("ocij,nctuij->notuij", U, V)                  # contraction over channels: fp32 accumulate
    Y = ATd @ M @ ATd.T                                            # N,O,th,tw,m,m
    th, tw = Y.shape[2], Y.shape[3]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(n, -1, th * m, tw * m)[:, :, :h, :wd]


def report(name, n, cin, cout, hw):
    x = F.leaky_relu(torch.randn(n, cin, hw, hw), 0.2)             # what a layer sees: the previous layer's LeakyReLU output
    w = torch.randn(cout, cin, 3, 3) * (2.0 / (cin * 9)) ** 0.5    # equalised-lr scale folded in, as the packed filters hold it
    ref = F.conv2d(x.double(), w.double(), padding=1)
    scale = ref.pow(2).mean().sqrt()
    rows = [("direct fp32", F.conv2d(x, w, padding=1).double()),
            ("F(2x2,3x3) fp32", wino(x, w, BT2, G2, AT2, torch.float32).double()),
            ("F(4x4,3x3) fp32", wino(x, w, BT4, G4, AT4, torch.float32).double()),
            ("F(4x4,3x3) fp64 (algebra check)", wino(x, w, BT4, G4, AT4, torch.float64))]
    print(f"{name}: {n} x {cin}->{cout} @ {hw}x{hw}, output rms {scale:.3f}")
    for label, y in rows:
        e = (y - ref) / scale
        print(f"   {label:34s} rms {e.pow(2).mean().sqrt():.3e}   max {e.abs().max():.3e}")


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    report("D2.0", n, 48, 64, 128)
    report("D3.0", n, 64, 80, 64)
    report("deep chain proxy (160 channels)", n, 160, 160, 32)
