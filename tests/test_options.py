"""rt_pose_amd.options.PlanOptions: the one place a plan's build-time choices live (and RTP_PLAN, the one environment variable)."""
import pytest

from rt_pose_amd.options import PlanOptions


def test_defaults_and_overrides(monkeypatch):
    o = PlanOptions()
    assert (o.merge_head, o.lazy_coef, o.no_tail, o.defer_wg, o.lanes, o.width_hints) == (1, 1, 0, "3", None, None)
    assert o.int_list("defer_wg") == [3] and o.int_list("lanes") is None and repr(o) == "PlanOptions()"
    o = PlanOptions(defer_wg="", lazy_coef=0)
    assert o.int_list("defer_wg") == [] and o.lazy_coef == 0 and "lazy_coef=0" in repr(o)
    with pytest.raises(KeyError):
        PlanOptions(no_such_field=1)
    monkeypatch.setenv("RTP_PLAN", "no_tail; lanes=0,1,2,3,4,5 ;width_hints=conv:s2.b0=160,wgrad:s3.b0=192; fused_fold=0")
    o = PlanOptions.from_env()
    assert o.no_tail == 1 and o.int_list("lanes") == [0, 1, 2, 3, 4, 5] and o.fused_fold == 0
    from rt_pose_amd.engine import parse_width_hints
    assert parse_width_hints(o.width_hints) == [("conv:s2.b0", 160), ("wgrad:s3.b0", 192)]
    monkeypatch.setenv("RTP_PLAN", "nonsense=1")
    with pytest.raises(KeyError):
        PlanOptions.from_env()


def test_the_product_reads_few_environment_variables():
    """at most 25 RTP_* switches in the product (Python + C), none of them read at import."""
    import glob
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    py = re.compile(r"""environ(?:\.get\(|\[)\s*["'](RTP_[A-Z0-9_]+)""")
    cc = re.compile(r"""getenv\("(RTP_[A-Z0-9_]+)""")
    for f in glob.glob(os.path.join(root, "rt_pose_amd", "*.py")) + [os.path.join(root, "bench.py")]:
        names |= set(py.findall(open(f).read()))
    for f in glob.glob(os.path.join(root, "rt_pose_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "rt_pose_amd", "csrc", "*.h")):
        names |= set(cc.findall(open(f).read()))
    assert len(names) <= 25, sorted(names)
