"""The box as a lone workgroup sees it (vc_box_probe) next to the one-sequence sampler's own stamps: python tools/box_probe.py [preset]
Prints one JSON line: dependent-load latency over a cache-resident and a 256 MB ring, the shader clock during those walks, the
sampler's in-kernel time (chip-wide 100 MHz counter) and its phase means in shader clocks."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda", 0)
box = bench.box_block(dev)
preset = sys.argv[1] if len(sys.argv) > 1 else "giga830M"
wl = bench.Workload(preset, "tts", 1, 80, 150, 40, "bf16", dev)
wl.call(1)
s1 = bench.sampler_block(wl, box)
box2 = bench.box_block(dev)
print(json.dumps({"preset": preset, "box_before": box, "sampler": s1, "box_after_two_calls": box2}), flush=True)
