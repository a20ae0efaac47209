"""Layer-l training steps four ways, ms per step: graph replay (SDAV.train_steps), eager steps with the masks drawn in front,
eager steps with the next step's masks drawn on a second stream."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
x = torch.rand((10, 30, 1681), dtype=torch.float64, device=eng.device)
for layer in (0, 2, 4):
    net = dlc.SDAV(seed=3, weight_scale="fan_in")
    with eng.latency_mode():
        net.train_steps(layer, x, 5); torch.cuda.synchronize()
        t0 = time.perf_counter(); net.train_steps(layer, x, 40); torch.cuda.synchronize()
        replay = (time.perf_counter() - t0) / 40 * 1e3
        sets = [[net._mask(l) for l in range(layer + 1)] for _ in range(2)]
        def fill(b):
            for l, m in enumerate(sets[b]):
                net._fill_mask(m, l)
        for _ in range(3):
            net.train_step(layer, x, sets[0])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(40):
            fill(0); net.train_step(layer, x, sets[0])
        torch.cuda.synchronize(); front = (time.perf_counter() - t0) / 40 * 1e3
        main, side = torch.cuda.current_stream(), eng.side_stream
        drawn, used = [torch.cuda.Event(), torch.cuda.Event()], [torch.cuda.Event(), torch.cuda.Event()]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fill(0); drawn[0].record(main)
        for i in range(40):
            b = i % 2
            main.wait_event(drawn[b])
            net.train_step(layer, x, sets[b])
            used[b].record(main)
            if i > 0: side.wait_event(used[1 - b])
            else: side.wait_stream(main)
            with torch.cuda.stream(side):
                fill(1 - b); drawn[1 - b].record(side)
        torch.cuda.synchronize(); beside = (time.perf_counter() - t0) / 40 * 1e3
    print("layer %d: graph replay %.3f, eager + masks in front %.3f, eager + masks beside %.3f ms per step" % (layer, replay, front, beside), flush=True)
