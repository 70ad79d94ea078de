"""what would the volume's consumer prefer?  The reference's encoder starts with a 3x3x3 Conv3d + BatchNorm +
ReLU and a 2x max-pool (models/regressor.py:70-71,78-80; Res3DBlock comes from the missing models.v2v, V2V-PoseNet style).
Times that first stage on MIOpen for the volume layouts / dtypes the un-projection kernel could emit, against the time the
un-projection itself takes to write them."""
import sys, time, torch, torch.nn as nn
dev = torch.device("cuda:0")
B, C, S, C1 = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 256, 64, 128
stage = nn.Sequential(nn.Conv3d(C, C1, 3, padding=1), nn.BatchNorm3d(C1), nn.ReLU(inplace=True), nn.MaxPool3d(2)).to(dev).eval()
print("first encoder stage on a (%d, %d, %d^3) volume: Conv3d(%d->%d, 3) + BN + ReLU + MaxPool(2), eval, MIOpen" % (B, C, S, C, C1))
for dt in (torch.float32, torch.float16, torch.bfloat16):
    for fmt, name in ((torch.contiguous_format, "(B,C,X,Y,Z)"), (torch.channels_last_3d, "channels_last_3d")):
        try:
            m = stage.to(dtype=dt).to(memory_format=fmt)
            x = torch.randn(B, C, S, S, S, device=dev, dtype=dt).contiguous(memory_format=fmt)
            with torch.no_grad():
                for _ in range(3): m(x)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(5): m(x)
                torch.cuda.synchronize()
            print("  %-9s %-18s %8.2f ms" % (str(dt).replace("torch.", ""), name, (time.perf_counter() - t0) / 5 * 1e3))
        except Exception as e:
            print("  %-9s %-18s failed: %s" % (str(dt).replace("torch.", ""), name, str(e)[:80]))
        del x
        torch.cuda.empty_cache()
