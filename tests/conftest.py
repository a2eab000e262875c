import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref built from /root/reference (build container)")


def pytest_collection_modifyitems(config, items):
    from backends import REF_BIN, REF_SO

    have_ref = os.path.exists(REF_BIN) and os.path.exists(REF_SO)
    skip_ref = pytest.mark.skip(reason="oracle/_ref (reference build) not present")
    for item in items:
        if "ref" in item.keywords and not have_ref:
            item.add_marker(skip_ref)
