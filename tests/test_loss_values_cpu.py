"""train_ops.LossValues on the host side: the numbers a training forward puts into tb_dict / disp_dict (reference ptt.py:44-60,
tools/train_utils/train_utils.py:55-70: `disp_dict.update(...)`, `tb_log.add_scalar('train/' + key, val, it)`) must behave as the
floats the reference puts there — without fetching anything until somebody looks."""
import numbers
import pickle

import numpy as np
import torch

from ptt_amd.train_ops import LossValues


def test_values_are_lazy_and_behave_as_floats():
    vals = LossValues(torch.tensor([9.0, 1.5, 2.5, 0.25, 4.0, 0, 0, 0]))
    a, b = vals[1], vals[2]
    assert vals.host is None                                            # nothing copied yet
    assert isinstance(a, numbers.Real) and np.isscalar(a)               # what tensorboard's make_np asks before np.array([x])
    assert vals.host is None
    assert float(a) == 1.5 and vals.host is not None                    # ONE copy serves all values
    assert a + b == 4.0 and b - a == 1.0 and 2 * a == 3.0 and b / a == 2.5 / 1.5 and a ** 2 == 2.25 and -a == -1.5 and abs(-a) == 1.5
    assert a < b and b >= a and a == 1.5 and a != b and max(a, b) is b and sum([a, b]) == 4.0
    assert "%.2f %s" % (a, b) == "1.50 2.5" and "{:.1f}".format(b) == "2.5" and repr(a) == "1.5" and round(b) == 2 and int(b) == 2
    assert np.array([a]).dtype == np.float64 and float(np.asarray(b)) == 2.5 and np.float32(a) == np.float32(1.5)
    assert pickle.loads(pickle.dumps(a)) == 1.5 and type(pickle.loads(pickle.dumps(a))) is float
    assert {a: "x"}[1.5] == "x"
    d = {}
    d.update({'loss_a': a, 'loss_b': b})
    assert {k: float(v) for k, v in d.items()} == {'loss_a': 1.5, 'loss_b': 2.5}
