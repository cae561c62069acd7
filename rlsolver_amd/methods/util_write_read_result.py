"""Result files in the reference's text format (rlsolver/methods/util_write_read_result.py:40-80):

    // obj: <value>
    // running_duration: <seconds>
    // num_nodes: <N>
    // alg_name: <name>
    <node id, 1-based> <label (+1 when plus1)>
"""
from __future__ import annotations

import os
from typing import Optional, Sequence, Union

import numpy as np


def write_graph_result(obj: Union[float, int], running_duration: Optional[int], num_nodes: Optional[int],
                       alg_name: str, solution: Sequence, filename: str, plus1: bool = True,
                       info_dict: Optional[dict] = None) -> str:
    sol = np.asarray(solution.detach().cpu().numpy() if hasattr(solution, "detach") else solution).astype(np.int64)
    with open(filename, "w", encoding="UTF-8") as f:
        f.write(f"// obj: {obj}\n")
        f.write(f"// running_duration: {running_duration}\n")
        if num_nodes is not None:
            f.write(f"// num_nodes: {num_nodes}\n")
        f.write(f"// alg_name: {alg_name}\n")
        for k, v in (info_dict or {}).items():
            f.write(f"// {k}: {v}\n")
        off = 1 if plus1 else 0
        f.write("".join(f"{i + 1} {int(s) + off}\n" for i, s in enumerate(sol)))
    return filename


def read_graph_result(filename: str):
    """-> (header dict, labels int64 array as stored)."""
    header, labels = {}, []
    with open(filename, "r", encoding="UTF-8") as f:
        for line in f:
            line = line.strip()
            if line.startswith("//"):
                k, _, v = line[2:].partition(":")
                header[k.strip()] = v.strip()
            elif line:
                labels.append(int(line.split()[1]))
    return header, np.asarray(labels, dtype=np.int64)
