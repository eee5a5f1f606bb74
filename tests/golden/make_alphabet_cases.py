#!/usr/bin/env python3
"""Extracts the DATA of the reference's alphabet test table
(test/test_charmodel_alphabet.c:36-347: thresholds, adjustments, flags and the
expected alphabet / collapse strings, for the erewhon.txt cases) into
tests/golden/alphabet_cases.json.  Run where /root/reference exists."""
import json, os, re
src = open('/root/reference/test/test_charmodel_alphabet.c').read()
cases = []
parts = src.split('.threshold')[1:]  # one chunk per table entry (strings may contain braces)
for part in parts:
    body = '.threshold' + part
    d = {}
    for k in ('threshold', 'digit_adjust', 'alpha_adjust', 'ignore_case', 'utf8', 'collapse_space'):
        mm = re.search(r'\.%s\s*=\s*([^,\n]+)' % k, body)
        d[k] = float(mm.group(1)) if mm else 0.0
    ok = True
    for k in ('alphabet', 'collapse'):
        mm = re.search(r'\.%s\s*=\s*((?:"(?:[^"\\]|\\.)*"\s*)+)' % k, body)
        if not mm:
            ok = False
            break
        s = ''.join(re.findall(r'"((?:[^"\\]|\\.)*)"', mm.group(1)))
        d[k] = re.sub(r'\\(.)', lambda m: {'n': '\n', 'r': '\r', 't': '\t'}.get(m.group(1), m.group(1)), s)
    mm = re.search(r'\.filename\s*=\s*(\w+)', body)
    if ok and mm and mm.group(1) == 'EREWHON_TEXT':
        cases.append(d)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'alphabet_cases.json')
json.dump(cases, open(out, 'w'), indent=1)
print(len(cases), 'cases ->', out)
