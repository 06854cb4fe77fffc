"""cProfile of the host side of eager DCGAN-64 steps (what an N > 1 run without capture pays per step).  gpurun: python scripts/probe/host_profile.py"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'ipr-gan_amd'))
import torch  # noqa: E402
import bench  # noqa: E402

sys.argv = ['bench.py', '--no-cpu-baseline', '--alt-math', 'none', '--graph', 'off', '--steps', '30', '--warmup', '10']
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(45)
st.sort_stats('cumulative').print_stats(60)
