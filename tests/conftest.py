import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "anemoi-rust_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

FIELD_IDS = ["bls12_381", "bls12_377", "bn_254", "ed_on_bls12_377", "jubjub", "pallas", "vesta"]
INSTANCES = [(f, w) for f in FIELD_IDS for w in (2, 4)]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """ANEMOI_TEST_ORDER=reverse | shuffle:<seed> runs the collected tests in another order (tests must not depend on what
    ran before them: constant tables, lanes and code objects are loaded on first use by whichever test comes first)."""
    order = os.environ.get("ANEMOI_TEST_ORDER", "")
    if order == "reverse":
        items.reverse()
    elif order.startswith("shuffle:"):
        import random
        random.Random(int(order.split(":", 1)[1])).shuffle(items)


class knobs:
    """Library options for the duration of a with-block, through anemoi_set_option (options are read from the
    environment once and changed only through the API; names are option names or their ANEMOI_* variables).
    None = the automatic default."""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        import anemoi_amd as A
        self.prev = {k: A.get_option(k) for k in self.kv}
        for k, v in self.kv.items():
            A.set_option(k, v)
        return self

    def __exit__(self, *exc):
        import anemoi_amd as A
        for k, v in self.prev.items():
            A.set_option(k, v)
        return False


def inst_key(field, width):
    return "%s/anemoi_%s" % (field, "2_1" if width == 2 else "4_3")


@pytest.fixture(scope="session")
def kats():
    with open(os.path.join(ROOT, "tests", "golden", "kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def params():
    with open(os.path.join(ROOT, "tests", "golden", "params.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    import orc
    orc.build()
    return orc.Oracle()
