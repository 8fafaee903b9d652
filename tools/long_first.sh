#!/bin/bash
for v in 0 32; do PDWT_SWT_NARROW=$v python3 tools/dbg_swt.py; done
