"""Observed parity errors of the GPU suite: every ``rel_err`` / ``note`` a GPU test evaluates is recorded per
test; ``conftest.pytest_sessionfinish`` writes the maxima to ``gpurun_out/parity_report.json`` (copied to ``profiles/`` per round), so the
margin between what is measured and what is asserted is on file.  Test infrastructure only."""
import os

import numpy as np

RECORDS = {}


def _test_id():
    return os.environ.get('PYTEST_CURRENT_TEST', 'outside-pytest').split(' (')[0]


def note(err, label=None, bound=None):
    """Record one observed error (a relative error unless ``label`` says otherwise) for the running test; returns it."""
    err = float(err)
    rec = RECORDS.setdefault(_test_id(), {'max_err': 0.0, 'checks': 0})
    rec['checks'] += 1
    if err >= rec['max_err'] or rec['checks'] == 1:
        rec['max_err'] = err
        if label is not None:
            rec['worst'] = str(label)
    if bound is not None:
        rec['bound'] = float(bound)
    return err


def rel_err(got, want, label=None):
    """max |got - want| / max |want| in float64 - the figure every parity assertion of the suite compares with its bound - recorded."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return note(np.abs(got - want).max() / max(np.abs(want).max(), 1e-30), label)


def write(path):
    import json
    out = {'what': 'largest error each GPU parity test observed (max |got - want| / max |want| unless the test names another figure); '
                   'bounds are in the tests', 'tests': {k: RECORDS[k] for k in sorted(RECORDS)}}
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
