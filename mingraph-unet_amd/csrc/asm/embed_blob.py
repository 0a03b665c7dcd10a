#!/usr/bin/env python3
"""embed_blob.py FILE SYMBOL OUT.cpp -- a binary file as `extern "C" const unsigned char SYMBOL[]` + `SYMBOL_len`."""
import sys

data = open(sys.argv[1], "rb").read()
sym = sys.argv[2]
with open(sys.argv[3], "w") as f:
    f.write(f'extern "C" {{\nextern const unsigned char {sym}[];\nextern const unsigned {sym}_len;\n')
    f.write(f"alignas(4096) const unsigned char {sym}[] = {{\n")
    for i in range(0, len(data), 32):
        f.write(",".join(str(b) for b in data[i:i + 32]) + ",\n")
    f.write(f"}};\nconst unsigned {sym}_len = {len(data)};\n}}\n")
