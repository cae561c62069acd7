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


def result_file_name(filename: str, running_duration: Optional[int]) -> str:
    """Where the reference would put the result of ``filename`` (util_write_read_result.py:55-67 with
    methods/util.py:200-211): a path under .../data/... goes to the same place under .../result/... (directory
    created), the running duration is appended as ``_<seconds>`` (``_`` alone when it is None); and an existing
    file is NEVER overwritten -- lowercase letters are appended until the name is free."""
    tail = "_" if running_duration is None else ("_" + str(int(running_duration)) if "data" in filename else None)
    new = filename.replace("data", "result") if "data" in filename else filename
    d = os.path.dirname(new)
    if d and not os.path.exists(d):
        os.makedirs(d, exist_ok=True)
    if tail is not None:
        new = new.replace(".txt", "") + tail + ".txt"
    rng = np.random.default_rng()
    while os.path.exists(new):
        stem, ext = os.path.splitext(new)
        new = stem + "abcdefghijklmnopqrstuvwxyz"[int(rng.integers(26))] + ext
    return new


def write_graph_result(obj: Union[float, int], running_duration: Optional[int], num_nodes: Optional[int],
                       alg_name: str, solution: Sequence, filename: str, plus1: bool = True,
                       info_dict: Optional[dict] = None) -> str:
    """Writes the result and returns the path actually used (see result_file_name: never an existing file)."""
    sol = np.asarray(solution.detach().cpu().numpy() if hasattr(solution, "detach") else solution).astype(np.int64)
    filename = result_file_name(filename, running_duration)
    with open(filename, "x", encoding="UTF-8") as f:
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


def write_result(obj: Union[float, int], running_duration: Optional[int], alg_name: str, solution: Sequence, filename: str,
                 plus1: bool = True, info_dict: Optional[dict] = None) -> str:
    """util_write_read_result.py:28-36: write_graph_result without the num_nodes line."""
    return write_graph_result(obj, running_duration, None, alg_name, solution, filename, plus1, info_dict)


def obtain_first_number(s: str) -> int:
    """The first run of digits / dots of ``s`` as an int (util_write_read_result.py:219-229)."""
    res, seen = "", False
    for ch in s:
        if ch.isdigit() or ch == ".":
            res += ch
            seen = True
        elif seen:
            break
    return int(float(res))


def read_graph_result_comments(filename: str):
    """-> (num_nodes, ID, running_duration, obj, obj_bound) from the ``//`` header of a result file whose name carries
    ``ID<i>_`` (util_write_read_result.py:139-158: reading stops at the num_nodes line).  obj_bound is None when the file has
    no such line (the reference leaves the name unbound there and raises)."""
    num_nodes = running_duration = obj = obj_bound = None
    ID = int(filename.split("ID")[1].split("_")[0])
    with open(filename, "r", encoding="UTF-8") as f:
        for line in f:
            if "//" not in line:
                continue
            if "num_nodes:" in line:
                num_nodes = int(line.split("num_nodes:")[1])
                break
            if "running_duration:" in line:
                running_duration = obtain_first_number(line)
            if "obj:" in line:
                obj = float(line.split("obj:")[1])
            if "obj_bound:" in line:
                obj_bound = float(line.split("obj_bound:")[1])
    return num_nodes, ID, running_duration, obj, obj_bound


def calc_obj_maxcut_xstr(x_str: str, filename: str, device=None) -> int:
    """Cut value of the base-64 solution string ``x_str`` on the graph file ``filename`` (util_write_read_result.py:232-239),
    evaluated by the HIP objective kernel."""
    import torch

    from ..envs.env_L2A import EnvMaxcut
    from ..graph import read_graph_header, read_mygraph
    from .util_evaluator import EncoderBase64
    mygraph = read_mygraph(filename)
    num_nodes = read_graph_header(filename)[0]        # read_nxgraph adds every node of the header line (util_read_data.py:46-66)
    dev = torch.device(device if device is not None else "cuda:0")
    env = EnvMaxcut(mygraph=mygraph, device=dev, num_nodes=num_nodes)
    x = EncoderBase64(encode_len=num_nodes).str_to_bool(x_str).to(dev)
    return int(env.calculate_obj_values(x[None, :])[0])


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
