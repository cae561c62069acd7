import os
"""Host-side helpers (CPU): base-64 solution strings, result files."""
import numpy as np
import pytest
import torch

from rlsolver_amd.methods.util_evaluator import EncoderBase64, Evaluator
from rlsolver_amd.methods.util_write_read_result import read_graph_result, write_graph_result


def test_encoder_matches_reference_golden(golden):
    z = golden("encoder_base64")
    for n in (5, 100, 800):
        enc = EncoderBase64(encode_len=n)
        x = torch.from_numpy(z[f"n{n}/x"].astype(bool))
        s = str(z[f"n{n}/str"])
        assert enc.bool_to_str(x) == s
        assert torch.equal(enc.str_to_bool(s), x)
        assert enc.string_len == -(-n // 6)
    for k in ("G14", "G15", "G22", "G49", "G50", "G55", "G70"):
        n = int(z[f"kat/{k}/num_nodes"])
        got = EncoderBase64(encode_len=n).str_to_bool(str(z[f"kat/{k}/str"]))
        assert np.array_equal(got.numpy().astype(np.uint8), z[f"kat/{k}/x"])


def test_encoder_edge_cases():
    for n in (1, 6, 7, 64, 65, 1000):
        enc = EncoderBase64(n)
        for x in (np.zeros(n, bool), np.ones(n, bool), np.arange(n) % 3 == 0):
            s = enc.bool_to_str(x)
            assert len(s.replace("\n", "")) >= enc.string_len or n > 384
            assert np.array_equal(enc.str_to_bool(s).numpy(), x)


def test_evaluator_has_no_cpu_path(tmp_path):
    xs = torch.tensor([[0, 1, 1], [1, 0, 0]], dtype=torch.bool)
    with pytest.raises(TypeError):
        Evaluator(str(tmp_path), 3, xs[0], 1.0, True)


def test_result_file_roundtrip(tmp_path):
    p = str(tmp_path / "res.txt")
    p = write_graph_result(13359, 12, 5, "dREINFORCE", torch.tensor([0, 1, 1, 0, 1]), p, plus1=True, info_dict={"seed": 3})
    txt = open(p).read().splitlines()
    assert txt[:5] == ["// obj: 13359", "// running_duration: 12", "// num_nodes: 5", "// alg_name: dREINFORCE",
                       "// seed: 3"]
    assert txt[5:] == ["1 1", "2 2", "3 2", "4 1", "5 2"]
    hdr, lab = read_graph_result(p)
    assert hdr["obj"] == "13359" and lab.tolist() == [1, 2, 2, 1, 2]


def test_result_file_is_never_overwritten(tmp_path):
    """util_write_read_result.py:55-67: data/ -> result/, `_<duration>` tail, a letter appended while the name exists."""
    os.makedirs(tmp_path / "data" / "syn")
    src = str(tmp_path / "data" / "syn" / "BA_100_ID0.txt")
    sol = torch.tensor([1, 0, 1])
    a = write_graph_result(7, 12, 3, "greedy", sol, src)
    b = write_graph_result(9, 12, 3, "greedy", sol, src)
    assert a == str(tmp_path / "result" / "syn" / "BA_100_ID0_12.txt") and os.path.exists(a)
    assert b != a and os.path.exists(b) and b.startswith(a[:-4]) and len(b) == len(a) + 1
    assert read_graph_result(a)[0]["obj"] == "7" and read_graph_result(b)[0]["obj"] == "9"
    c = write_graph_result(1, None, 3, "greedy", sol, str(tmp_path / "plain.txt"))
    assert c.endswith("plain_.txt")
