"""reinfocus_amd -- MI355X-native render-and-measure path of reinfocus.

Only what the hot path needs lives here: the C-ABI HIP library (csrc/), its ctypes
binding (_native), and host-side mirrors of the reference interface for that path
(graphics.render.FastRenderer, vision.focus_values, environments.FocusObserver and
the DiscreteSteps-v0 vector environment).
"""

__version__ = "0.1.0"
