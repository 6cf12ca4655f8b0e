"""`ptt` import alias: the reference's tools import `ptt.config`, `ptt.models...` (tools/train_tracking.py:12-14,
tools/test_tracking.py). With this repository on PYTHONPATH those imports resolve to the MI355X implementation in
ptt_amd/ (same module paths, names and signatures). The dataset / IO / utility packages of the reference
(`ptt.datasets`, `ptt.utils`) are outside the hot path and are NOT provided: keep the
reference's own copies of those two directories next to this alias if you run its tools end to end."""
import os

__path__ = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ptt_amd")]
__version__ = "0.1.0+mi355x"
