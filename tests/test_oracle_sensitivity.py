"""What the reference's 304 golden lines pin -- and what they do not.

The oracle restates several OpenCV 3.4.5 semantics from knowledge of its sources (OpenCV is absent: SURVEY.md
appendix A).  tools/oracle_sensitivity.py flips each one and counts the golden lines that notice; the table is
committed (tests/golden/oracle_sensitivity.json) and re-derived here, so that it cannot go
stale and so that nobody claims more than the goldens hold: a semantic with 0 changed lines is pinned by nothing in
the reference -- for it the HIP path equals the oracle, and that is all that can be said."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def test_sensitivity_table_is_current():
    import oracle_sensitivity as osens
    rows = osens.audit()
    with open(os.path.join(ROOT, 'tests', 'golden', 'oracle_sensitivity.json')) as fp:
        committed = json.load(fp)
    assert rows == committed
    by = {r['switch']: r for r in rows}
    # the goldens DO pin the float32 HLS path (rounding mode, float L) and the hole-filling drawContours ...
    assert by['hls_round=1']['golden_lines_changed'] > 0
    assert by['l_integer=1']['golden_lines_changed'] > 0
    assert by['no_hole_fill=1']['golden_lines_changed'] > 0
    # ... and do NOT pin these: no fixture exercises them (documented as unpinned in DESIGN.md section 2)
    for k in ('hls_variant=1', 'hls_variant=2', 'contour_tie=1', 'mean_form=1', 'area_rule=1', 'erode_border=1', 'hue_g_first=1'):
        assert by[k]['golden_lines_changed'] == 0 and by[k]['records_changed'] == 0, k
    # after the audit the oracle is back to the restatement proper
    from oracle import pyoracle as po
    import glob
    p = po.Params(os.path.join(ROOT, 'tests', 'golden', 'sample-images1', 'params.yml'))
    f = os.path.join(ROOT, 'tests', 'golden', 'sample-images1', '20180814215230-01-e136.jpg')
    assert po.run_file(f, p, display_name='e136')[0] == 'e136: 253.623'
