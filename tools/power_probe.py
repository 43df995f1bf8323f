"""Is the forward power-limited?  Replays the captured cfg-2 bf16 forward for a few seconds while a thread samples the GPU's
power / shader-clock sensors (sysfs hwmon, rocm-smi fallback), for (a) the synthetic N(0,1) feature maps of the benchmark and
(b) all-zero feature maps (same instruction stream per kernel, almost no operand toggling).  GPU only.
  tools/power_probe.py [seconds]"""
import glob, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mvgformer_amd.decoder import DecoderContext
from mvgformer_amd.factory import build_decoder_for_case, case_to_device
from mvgformer_amd.synthetic import build_case

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
dev = torch.device("cuda", 0)


def sensors():
    out = {}
    for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name in ("power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input", "temp2_input"):
            f = os.path.join(h, name)
            if os.path.exists(f):
                try:
                    out[name] = int(open(f).read().strip())
                except Exception:
                    pass
    return out


def smi():
    try:
        return subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True,
                              timeout=20).stdout
    except Exception as e:
        return "rocm-smi: %s" % e


print("sensors available:", sensors())
print(smi()[:1500])
case = build_case("cfg2", B=1, seed=0)
dec = build_decoder_for_case(case, dev, torch.bfloat16)
g = case_to_device(case, dev)
ctx = DecoderContext.prepare(g.spatial_shapes, g.level_start_index, g.meta, case.img_size, torch.bfloat16, 1, dev)


def run(label, src_views):
    def forward():
        ctx.feat = None
        return dec(g.tgt, g.reference_points, src_views, g.meta, g.spatial_shapes, g.level_start_index, None, query_pos=g.query_pos,
                   threshold=0.1, context=ctx)
    with torch.no_grad():
        for _ in range(3):
            forward()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            forward()
        graph.replay(); torch.cuda.synchronize()
        samples, stop = [], threading.Event()

        def poll():
            while not stop.is_set():
                samples.append((time.perf_counter(), sensors()))
                time.sleep(0.02)
        th = threading.Thread(target=poll); th.start()
        t0 = time.perf_counter(); n = 0
        marks = []
        while time.perf_counter() - t0 < secs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                graph.replay()
            e1.record(); torch.cuda.synchronize()
            marks.append(e0.elapsed_time(e1) / 100)
            n += 100
        stop.set(); th.join()
    keys = sorted({k for _, s in samples for k in s})
    print("== %s: %d forwards, ms per forward first / median / last block of 100: %.4f / %.4f / %.4f" % (
        label, n, marks[0], sorted(marks)[len(marks) // 2], marks[-1]))
    half = samples[len(samples) // 2:]
    for k in keys:
        v = [s[k] for _, s in half if k in s]
        if v:
            print("   %-16s second half of the run: mean %.1f  min %d  max %d" % (k, sum(v) / len(v), min(v), max(v)))
    print(smi()[:900])


run("N(0,1) feature maps (the benchmark's)", g.src_views)
run("all-zero feature maps", [torch.zeros_like(s) for s in g.src_views])
run("N(0,1) again", g.src_views)
