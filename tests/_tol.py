"""Recorder of the pinned tolerances: every parity assertion that goes through ``within`` leaves (case, tensor class, bar,
observed) behind; tests/conftest.py writes them to gpurun_out/tolerances.json at the end of the session (a copy of the GPU
run is kept under profiles/, and DESIGN.md 4 tabulates it)."""
RECORDS = []


def within(case, cls, observed, bar, what=''):
    """Records the observation and returns ``observed <= bar`` (use inside an assert)."""
    RECORDS.append({'case': str(case), 'class': cls, 'observed': float(observed), 'bar': float(bar), 'what': what})
    return float(observed) <= float(bar)


def summary():
    out = {}
    for r in RECORDS:
        e = out.setdefault(r['class'], {'bar': r['bar'], 'worst_observed': 0.0, 'worst_case': None, 'n': 0, 'what': r['what']})
        e['n'] += 1
        e['bar'] = max(e['bar'], r['bar'])
        if r['observed'] >= e['worst_observed']:
            e['worst_observed'], e['worst_case'] = r['observed'], r['case']
    return out
