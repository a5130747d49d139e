#!/usr/bin/env python3
"""Extract the reference's own known-answer values for the kriging hot path.

Run ONCE in the build container (where /root/reference exists); the outputs
(`golden_a.json`, `golden_b.json`, `kat.json`) are committed and are what
travels to the GPU box.  Nothing here is reference *source*: the notebook
values are executed cell outputs (data), the KAT arrays are the numeric
literals the reference's unit tests assert against (test data).

  golden_b : doc/Gpx_Tutorial.ipynb cell 31 output (line ~421): full-precision
             serde JSON of a fitted Linear-mean + Matern-5/2 GP (n=6, d=1)
  golden_a : doc/Gpx_Tutorial.ipynb cell 14 output (lines ~165-167): Constant
             mean + squared exponential (n=5), theta printed to 8 digits
  kat      : numeric literals of crates/gp/src/utils.rs:151-242,
             crates/gp/src/correlation_models.rs:598-641,719-726,
             crates/gp/src/mean_models.rs:170-178,196-214, crates/gp/src/algorithm.rs:1723-1797 and
             python/egobox/tests/test_gpmix.py:37-53
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def nd(obj):
    """serde-ndarray {"v":1,"dim":[..],"data":[..]} -> nested list."""
    import numpy as np
    return np.asarray(obj["data"], dtype=float).reshape(obj["dim"]).tolist()


def main():
    nb = json.load(open(os.path.join(REF, "doc", "Gpx_Tutorial.ipynb")))
    cells = nb["cells"]

    def stream(i):
        return "".join("".join(o["text"]) for o in cells[i].get("outputs", [])
                       if o.get("output_type") == "stream")

    # ---- golden B -------------------------------------------------------
    txt = stream(31)
    start = txt.index('{"recombination"')
    obj, _ = json.JSONDecoder().raw_decode(txt[start:])
    e = obj["experts"][0]
    ip = e["inner_params"]
    gb = {
        "source": "doc/Gpx_Tutorial.ipynb cell 31 (egobox 0.32.0), expert 0 of Gpx repr",
        "type_fullgp": e["type_fullgp"],
        "mean": "Linear", "corr": "Matern52",
        "theta": nd(e["theta"]),
        "likelihood": e["likelihood"],
        "sigma2": ip["sigma2"],
        "beta": nd(ip["beta"]), "gamma": nd(ip["gamma"]),
        "r_chol": nd(ip["r_chol"]), "ft": nd(ip["ft"]), "ft_qr_r": nd(ip["ft_qr_r"]),
        "w_star": nd(e["w_star"]),
        "xt_norm": {k: nd(v) for k, v in e["xt_norm"].items()},
        "yt_norm": {k: nd(v) for k, v in e["yt_norm"].items()},
        "training_x": nd(e["training_data"][0]), "training_y": nd(e["training_data"][1]),
        "nugget": e["params"]["nugget"],
        "params": {k: v for k, v in e["params"].items() if k in ("kpls_dim", "n_start", "max_eval")},
    }
    json.dump(gb, open(os.path.join(HERE, "golden_b.json"), "w"), indent=1)

    # ---- golden A -------------------------------------------------------
    txt = stream(14)
    th = float(re.search(r"Optimal theta = \[([0-9.eE+-]+)\]", txt).group(1))
    var = float(re.search(r"GP variance = ([0-9.eE+-]+)", txt).group(1))
    lk = float(re.search(r"Reduced likelihood = ([0-9.eE+-]+)", txt).group(1))
    ga = {
        "source": "doc/Gpx_Tutorial.ipynb cells 9,11,14 (egobox 0.32.0)",
        "mean": "Constant", "corr": "SquaredExponential",
        "xt": [0.0, 1.0, 2.0, 3.0, 4.0], "yt": [0.0, 1.0, 1.5, 0.9, 1.0],
        "theta_printed_8_digits": th, "variance": var, "likelihood": lk,
    }
    json.dump(ga, open(os.path.join(HERE, "golden_a.json"), "w"), indent=1)

    # ---- unit-test KATs (numeric literals of the reference's tests) -----
    kat = {
        "pairwise_differences": {  # crates/gp/src/utils.rs:151-176
            "x": [[-0.9486833], [-0.82219219]],
            "y": [[-1.26491106], [-0.63245553], [0.0], [0.63245553], [1.26491106]],
            "expected": [[0.31622777], [-0.31622777], [-0.9486833], [-1.58113883], [-2.21359436],
                         [0.44271887], [-0.18973666], [-0.82219219], [-1.45464772], [-2.08710326]],
            "tol": 1e-6},
        "normalize": {  # utils.rs:202-208
            "x": [[1.0, 2.0], [3.0, 4.0]], "mean": [2.0, 3.0], "std_sq": [2.0, 2.0]},
        "diff_matrix": {  # utils.rs:211-242
            "xt": [[0.5], [1.2], [2.0], [3.0], [4.0]],
            "d": [[0.7], [1.5], [2.5], [3.5], [0.8], [1.8], [2.8], [1.0], [2.0], [1.0]],
            "idx": [[0, 1], [0, 2], [0, 3], [0, 4], [1, 2], [1, 3], [1, 4], [2, 3], [2, 4], [3, 4]]},
        "sqexp_1d": {  # correlation_models.rs:598-616
            "xt": [[4.5], [1.2], [2.0], [3.0], [4.0]], "theta_sq": [0.2], "w": [[1.0]],
            "expected": [0.336552878364737, 0.5352614285189903, 0.7985162187593771,
                         0.9753099120283326, 0.9380049995307295, 0.7232502423798424,
                         0.4565760496233148, 0.9048374180359595, 0.6703200460356393,
                         0.9048374180359595], "tol": 1e-6},
        "sqexp_2d": {  # :619-630
            "xt": [[0.0, 1.0], [2.0, 3.0], [4.0, 5.0]], "theta_sq": [2.0, 4.0],
            "expected": [6.14421235e-06, 1.42516408e-21, 6.14421235e-06], "tol": 1e-6},
        "matern32_2d": {  # :633-641
            "xt": [[0.0, 1.0], [2.0, 3.0], [4.0, 5.0]], "theta": [1.0, 2.0],
            "expected": [1.08539595e-03, 1.10776401e-07, 1.08539595e-03], "tol": 1e-6},
        "matern52_2d": {  # :719-726
            "xt": [[0.0, 1.0], [2.0, 3.0], [4.0, 5.0]], "theta": [1.0, 2.0],
            "expected": [6.62391590e-04, 1.02117882e-08, 6.62391590e-04], "tol": 1e-6},
        "quadratic": {  # mean_models.rs:170-178
            "x": [[1.0, 2.0, 3.0], [3.0, 4.0, 5.0]],
            "expected": [[1.0, 1.0, 2.0, 3.0, 1.0, 2.0, 3.0, 4.0, 6.0, 9.0],
                         [1.0, 3.0, 4.0, 5.0, 9.0, 12.0, 15.0, 16.0, 20.0, 25.0]]},
        "quadratic2": {  # mean_models.rs:181-186
            "x": [[0.0], [7.0], [25.0]],
            "expected": [[1.0, 0.0, 0.0], [1.0, 7.0, 49.0], [1.0, 25.0, 625.0]]},
        "quadratic_jac": {  # mean_models.rs:196-214
            "x": [1.0, 2.0, 3.0],
            "expected": [[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [2.0, 0.0, 0.0],
                         [2.0, 1.0, 0.0], [3.0, 0.0, 1.0], [0.0, 4.0, 0.0], [0.0, 3.0, 2.0], [0.0, 0.0, 6.0]]},
        "bug_var_derivatives": {  # algorithm.rs:1723-1797: fixed data set + fixed theta, d var / dx vs central differences
            "xt": [[6.875, -4.375], [-3.125, 1.875], [1.875, -1.875], [-4.375, 3.125], [8.125, 9.375],
                   [4.375, 4.375], [0.625, 0.625], [9.375, 6.875], [5.625, 8.125], [-0.625, -3.125],
                   [3.125, 5.625], [-1.875, -0.625]],
            "yt": [2.43286801, 13.10840811, 5.32908578, 17.81862219, 74.08849877, 39.68137781, 14.96009727,
                   63.17475741, 61.26331775, -7.46009727, 44.39159189, 2.17091422],
            "theta_sq_half": [0.0437386, 0.00697978],  # theta = sqrt(2 * .)
            "x": [-1.3, 2.5], "e": 5e-6, "epsilon": 1e-5},
        "griewank": {  # crates/gp/src/algorithm.rs:1319-1323 (the synthetic workload's function, raw inputs)
            "x": [[1.0, 1.0, 1.0, 1.0, 1.0], [2.0, 2.0, 2.0, 2.0, 2.0]], "expected": [0.72890641, 1.01387135], "tol": 1e-8},
        "python_kriging": {  # python/egobox/tests/test_gpmix.py:24-53 (default fit => theta* of golden A)
            "xt": [0.0, 1.0, 2.0, 3.0, 4.0], "yt": [0.0, 1.0, 1.5, 0.9, 1.0],
            "predict_1.0": 1.0, "var_1.0": 0.0, "places": 7,
            "predict_1.1": 1.1163, "var_1.1": 0.0, "delta": 1e-3,
            "predict_gradients_1.1": 1.1204, "predict_var_gradients_1.1": 0.0145},  # test_gpmix.py:47-52
    }
    json.dump(kat, open(os.path.join(HERE, "kat.json"), "w"), indent=1)
    print("wrote golden_a.json golden_b.json kat.json")


if __name__ == "__main__":
    main()
