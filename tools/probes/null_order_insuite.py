"""tools/probes/null_stream_order_probe.py inside a pytest process that has run the GPU suite before it -- the state GPUTEST_r05
failed in (dozens of streams made, several contexts alive, the library's own pipelines warm).  Not collected by default:
    SS_TEST_ORDER=collection python -m pytest tests/test_gpu_docs.py tests/test_gpu_intermediates.py tests/test_gpu_kats.py \
        tests/test_gpu_minimal.py tests/test_gpu_parity.py tools/probes/null_order_insuite.py -m gpu -q -s
(round 5's collection order up to the test that failed)."""
import os

import pytest

from null_stream_order_probe import measure
from stark_symphony_amd import verifier

pytestmark = pytest.mark.gpu


def test_where_the_null_stream_fill_lands_after_the_suite():
    ver = verifier.Verifier(0)
    for join in (False, True, False):
        where, stale, late = measure(ver, 100, join)
        print("\nin-suite GPU_MAX_HW_QUEUES=%s join=%s reps=100 slots=5: fill event %s; status ended 0x55: %d of 500; fill later than the last pass by (ms): %s" % (
            os.environ.get("GPU_MAX_HW_QUEUES", "unset"), join, where, stale, [round(x, 3) for x in late[:8]]), flush=True)
        if join:
            assert stale == 0 and where["before_first"] == 500
