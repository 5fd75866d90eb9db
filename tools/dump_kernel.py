#!/usr/bin/env python3
"""Disassembly of one kernel of a HIP library: python tools/dump_kernel.py <lib.so> <kernel-name-regex> > out.s"""
import re
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_mix

text = isa_mix.disassemble(sys.argv[1])
for m in re.finditer(r"^[0-9a-f]+ <(\S+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", text, re.S | re.M):
    name = isa_mix.demangle(m.group(1))
    if re.search(sys.argv[2], name):
        print("; ----", name)
        print(m.group(2))
