"""WGSL include expansion as the reference does it (src/shaders/preprocessor.rs:6-23).

Every `#include NAME` line of the named shader is replaced, newline included, by the text of
NAME from the same directory.  Only the includes of the top file are expanded (the reference
iterates over the matches of the unexpanded source), and every occurrence of one include line
is replaced at once (`String::replace`).  Used by the test that checks the shader constants
against the device arithmetic on the shaders as the reference would compile them.
"""
from __future__ import annotations

import os
import re

_INCLUDE = re.compile(r"#include (.*?)\n")


def preprocess_wgsl(shader_dir: str, filename: str) -> str:
    with open(os.path.join(shader_dir, filename), "r", encoding="utf-8") as f:
        source = f.read()
    original = source
    for m in _INCLUDE.finditer(original):
        with open(os.path.join(shader_dir, m.group(1)), "r", encoding="utf-8") as f:
            nested = f.read()
        source = source.replace(m.group(0), nested)
    return source
