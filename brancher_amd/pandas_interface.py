"""
The pandas edge of the API (`brancher/pandas_interface.py:8-58`): the wire format of ``get_sample`` /
``get_posterior_sample`` (one DataFrame row per Monte-Carlo sample, one column per variable) and of
``observe(DataFrame)`` (one row per datapoint).  Host-side only: the samples come from the device as tensors in the
reference layout [N, B, d1, d2] (`engine._run_sampler`) and are unpacked here; a frame handed to ``observe`` becomes the
[datapoints, ...] arrays the lowering stages into the observed-data buffer.
"""
import numpy as np
import pandas as pd


def _cell(sample_value):
    """what one sample of one variable looks like in a frame: a float for a scalar, the vector for a single datapoint
    with several components, otherwise the array with its datapoint axis first (`pandas_interface.py:33-43`)"""
    if isinstance(sample_value, dict):               # a deterministic node that holds a network's dict of outputs
        return {key: _cell(value) for key, value in sample_value.items()}
    a = np.asarray(sample_value)
    if a.size == 1:
        return float(a.reshape(-1)[0])
    if a.shape[0] == 1:
        return a[0]
    return a


def reformat_sample_to_pandas(sample):
    """{variable: array-like [N, ...]} -> DataFrame with N rows and a column per (non-root) variable name"""
    columns = {}
    for var, value in sample.items():
        if type(var).__name__ == "RootVariable":
            continue
        if isinstance(value, dict):
            parts = {key: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)) for key, v in value.items()}
            n_rows = next(iter(parts.values())).shape[0]
            columns[var.name] = [_cell({key: a[n] for key, a in parts.items()}) for n in range(n_rows)]
            continue
        arr = value.detach().cpu().numpy() if hasattr(value, "detach") else np.asarray(value)
        columns[var.name] = [_cell(arr[n]) for n in range(arr.shape[0])]
    return pd.DataFrame(columns)


def pandas_frame2value(data, index):
    """the column `index` of a frame as ONE array whose leading axis runs over the rows (datapoints); anything that is
    not a frame passes through (`pandas_interface.py:24-30`)"""
    if not isinstance(data, pd.DataFrame):
        return data
    return np.array([np.asarray(cell).tolist() for cell in data[index].values])


def pandas_frame2dict(data):
    """{column: array over rows}; dictionaries pass through (`pandas_interface.py:15-21`)"""
    if isinstance(data, dict):
        return data
    if not isinstance(data, pd.DataFrame):
        raise ValueError("expected a dictionary or a pandas DataFrame, got {}".format(type(data).__name__))
    return {column: pandas_frame2value(data.sort_index(), column) for column in data}


def reformat_model_summary(summary_data, var_names, feature_list):
    return pd.DataFrame(summary_data, index=var_names, columns=feature_list).transpose()
