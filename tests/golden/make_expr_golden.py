#!/usr/bin/env python3
"""Golden vectors for the filter-expression language: each expression is parsed and evaluated by the
COMPILED reference (oracle/_ref/libbgt_ref.so: ke_parse / ke_set_int / ke_eval) -> tests/golden/expr.json."""
import ctypes as C
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ref = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "libbgt_ref.so"))
ref.ke_parse.restype = C.c_void_p
ref.ke_parse.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
ref.ke_set_int.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
ref.ke_eval.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_char_p), C.POINTER(C.c_int)]
ref.ke_destroy.argtypes = [C.c_void_p]

EXPRS = ["AC>0", "AC>=1&&AN>10", "AC1>0&&AC2==0", "AC1/AN1>=0.1&&AC2==0", "AC/AN", "AC//AN", "AC%7", "AC*2+AN", "2**3**2",
         "-AC+3", "+AC", "!AC", "~AC", "(AC+1)*(AN-1)", "AC<<2", "AN>>1", "AC&3", "AC|8", "AC^5", "AC==AN", "AC!=AN",
         "AC<>AN", "AC<AN", "AC<=AN", "AC>AN", "AC>=AN", "AC&&AN", "AC||AN", "1.5*AC", "AC/2", "AC/2==2", "AC/2>1.9",
         "10", "0x10+010", "1e2", ".5+AC", "abs(-3)", "abs(AC-AN)", "abs(-2.5)", "AC3>0", "AC+", "(AC", "AC)", "AC 1",
         "AC>0 && AN > 3", "AN-AC*2", "AC*-1", "2*(-1)", "AN/3*3", "AN//3*3", "7//2", "-7//2", "7%3", "AC2+AN2", "\"a\"==\"a\"",
         "\"a\"<\"b\"", "'x'=='y'", "AC>0||", "&&AC", "foo(AC)", "AC,AN", "3.0==3", "AC**0.5", "AC/AN>=0.05&&AC/AN<=0.95",
         "AN1+AN2==AN", "1<<40", "AC>1e-3", "2+3*4", "(2+3)*4", "10-3-2", "2**-1", "!0", "!!AN", "~0"]
VARS = [{"AC": 5, "AN": 20, "AC1": 3, "AN1": 10, "AC2": 0, "AN2": 10}, {"AC": 0, "AN": 7, "AC1": 0, "AN1": 3, "AC2": 4, "AN2": 4},
        {"AC": 4, "AN": 4, "AC1": 1, "AN1": 2, "AC2": 0, "AN2": 2}]
out = []
for e in EXPRS:
    err = C.c_int(0)
    ke = ref.ke_parse(e.encode(), C.byref(err))
    item = {"expr": e, "parse_err": err.value, "eval": []}
    if ke:
        for v in VARS:
            for k, x in v.items():
                ref.ke_set_int(ke, k.encode(), x)
            i, r, s, t = C.c_int64(0), C.c_double(0), C.c_char_p(), C.c_int(0)
            ee = ref.ke_eval(ke, C.byref(i), C.byref(r), C.byref(s), C.byref(t))
            item["eval"].append({"err": ee, "type": t.value, "i": i.value, "r": repr(r.value)})
        ref.ke_destroy(ke)
    out.append(item)
json.dump({"vars": VARS, "cases": out}, open(os.path.join(HERE, "expr.json"), "w"), indent=0)
print(len(out), "expressions")
