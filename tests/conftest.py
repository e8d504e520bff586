import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def example_gfa():
    with open(os.path.join(ROOT, "tests", "golden", "example_graph.gfa")) as f:
        return f.read()


@pytest.fixture(scope="session")
def example_reads():
    names, reads = [], []
    with open(os.path.join(ROOT, "tests", "golden", "example_reads.fa")) as f:
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                names.append(line[1:])
            elif line:
                reads.append(line)
    return names, reads
