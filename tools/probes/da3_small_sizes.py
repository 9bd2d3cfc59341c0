import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import gpu_diag as G
from burn_depth_amd.config import DepthAnything3Config, Precision
from burn_depth_amd.depth_pro import Device
dev = Device(0)
for size, w in ((1036, 0), (266, 518)):
    for prec in (Precision.F32, Precision.BF16, Precision.F16X2):
        cfg = DepthAnything3Config.small()
        cfg.image_size, cfg.image_width = size, w
        G.run_da3(dev, cfg, f"da3-small-{size}x{w or size}/p{int(prec)}", 1, prec, f16_weights=True)
bad = [r for r in G.RESULTS if not r[3]]
print("BAD:", bad)
sys.exit(1 if bad else 0)
