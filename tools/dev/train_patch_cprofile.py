#!/usr/bin/env python3
"""Development aid: cProfile of the host side of the training step at patch size (where does the Python time go?)."""
import cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_train_patch as B

sys.argv = ["bench_train_patch.py"] + sys.argv[1:]
B.main()            # warm-up pass in the same process (MIOpen searches, packed weights, allocator)
pr = cProfile.Profile()
pr.enable()
B.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(60)
print(s.getvalue()[:6000])
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.print_callers("torch.tensor}")
st.print_callers("'nonzero'")
st.print_callers("torch.rand}")
st.print_callers("'item'")
st.print_callers("'tolist'")
st.print_callers("'cpu'")
print(s.getvalue()[:8000])
