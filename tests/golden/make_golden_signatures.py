"""Fixture G12: the positional signatures of the reference's native modules, read from the
`PyArg_ParseTuple` format strings and the `PyMethodDef` tables of /root/reference/src_c/*.c
(module -> function -> required / optional argument counts and the format string).  Data only:
a few dozen short strings; run in the build container, where the reference sources exist.

    python tests/golden/make_golden_signatures.py
"""
import glob
import json
import os
import re

REF = '/root/reference/src_c'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'g12_signatures.json')


def main():
    out = {}
    for path in sorted(glob.glob(os.path.join(REF, '*.c'))):
        src = open(path).read()
        module = os.path.splitext(os.path.basename(path))[0]
        table = re.search(r'static\s+PyMethodDef\s+\w+\[\]\s*=\s*\{(.*?)\{\s*NULL', src, re.S)
        if not table:
            continue
        methods = re.findall(r'\{\s*"(\w+)"\s*,\s*(\w+)\s*,\s*(\w+)', table.group(1))
        funcs = {}
        for pyname, cname, flags in methods:
            body = re.search(r'static\s+PyObject\s*\*\s*' + cname + r'\s*\(.*?\n\}', src, re.S)
            fmt = re.search(r'PyArg_ParseTuple\s*\(\s*args\s*,\s*"([^"]*)"', body.group(0)) \
                if body else None
            if not fmt:
                continue
            f = fmt.group(1)
            req, _, opt = f.partition('|')
            funcs[pyname] = {'format': f, 'required': len(req), 'optional': len(opt),
                             'flags': flags}
        if funcs:
            out[module] = funcs
    json.dump(out, open(OUT, 'w'), indent=1, sort_keys=True)
    print(OUT, {m: len(f) for m, f in out.items()})


if __name__ == '__main__':
    main()
