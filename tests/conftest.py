import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


DEV_SWITCH_DEFAULTS = {"attn_variant": 0, "attn_force_split": 0, "s6_prio": 0, "s6_early_out": 3, "decode_attn_valu": 0, "attn_debug": 0, "attn_flat": -1}


@pytest.fixture
def dev_switch():
    """Set a developer A/B switch of the library for this test (qp_dev_switch: process-wide atomics — the launch paths do not read the
    environment); every switch touched goes back to its default afterwards."""
    from quickvideo_amd import native
    lib = native.load_library()
    touched = set()

    def set_(name, value):
        touched.add(name)
        assert lib.qp_dev_switch(name.encode(), int(value)) == 0, lib.qp_last_error().decode()

    yield set_
    for name in touched:
        lib.qp_dev_switch(name.encode(), DEV_SWITCH_DEFAULTS[name])
