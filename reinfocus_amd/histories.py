"""Per-environment event histories (reference: reinfocus/histories.py:10-83).

`Histories(num_histories, max_n)` keeps the last `max_n` float32 events of every environment;
empty positions are NaN, the newest event is the last column.  Host-side bookkeeping for
the visualiser -- nothing here touches the GPU.
"""

import numpy as np


class Histories:
    def __init__(self, num_histories, max_n):
        self.data = np.full((num_histories, max_n), np.nan, dtype=np.float32)

    def get_history(self, index):
        """The recorded (non-NaN) events of one history, oldest first (histories.py:22-33)."""
        row = self.data[index]
        return row[~np.isnan(row)]

    def most_recent_events(self):
        """The newest event of every history, NaN where there is none (histories.py:35-41)."""
        return self.data[:, -1]

    def append_events(self, events, indices=None):
        """Shifts the selected histories left by one and stores one new event in each
        (histories.py:43-66); `events` has one entry per selected history."""
        selected = slice(None) if indices is None else np.asarray(indices, dtype=bool)
        kept = self.data[selected, 1:]
        new = np.asarray(events, dtype=np.float32).reshape(len(kept), 1)
        self.data[selected] = np.concatenate([kept, new], axis=1)

    def reset(self, indices):
        """Empties the selected histories (histories.py:68-79)."""
        self.data[np.asarray(indices, dtype=bool)] = np.nan
