"""The launches of rank 0's plan for the Llama-3-8B sweep at a world size: kind, layers, bytes, selection form."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
for world in [int(a) for a in sys.argv[1:]] or [8]:
    wl = bench.ModelSweep("llama8b", torch.device("cuda", 0), 0, world, "normal")
    print("world", world, "families", len(wl.plan.families))
    for ps in wl.plan.plan:
        for la in ps["launches"]:
            jobs = [x for x in la if isinstance(x, list)]
            print("  ", la[0], [x for x in la[1:] if isinstance(x, (int, float))][:6],
                  "jobs", [(j.get("bytes"), j.get("split")) for js in jobs for j in js if isinstance(j, dict)][:4])
