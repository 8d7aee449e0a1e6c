"""Minimal pandas edge (`brancher/pandas_interface.py`): host-side convenience only."""
import numpy as np
import pandas as pd


def pandas_frame2value(data, index):
    if isinstance(data, pd.DataFrame):
        return np.array([np.asarray(v) for v in data[index].values])
    return data


def reformat_model_summary(summary_data, var_names, feature_list):
    return pd.DataFrame(summary_data, index=var_names, columns=feature_list).transpose()
