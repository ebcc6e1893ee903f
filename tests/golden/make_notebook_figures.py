"""Extracts the figures the reference's notebooks hold as OUTPUTS -- images the real reference
(numba on CUDA, OpenCV 4.9, matplotlib 3.8) rendered -- into tests/golden/notebook_figures/:

  examples/render.ipynb       cell 3: five scenes through render.render (the general renderer:
                              rectangles, spheres, rotated cameras), one pyplot.imshow figure each
  examples/environment.ipynb  cells 6, 9, 12, 13, 14: eight e.render() figures of one
                              DiscreteSteps episode (600 px FastRenderer frame | performance plot)

Only output data is copied (the PNG bytes as stored in the notebooks); no source.  Runs where
/root/reference exists:  python tests/golden/make_notebook_figures.py
"""

import base64
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference/examples"


def figures_of(notebook, cells):
    cells_json = json.load(open(os.path.join(REFERENCE, notebook)))["cells"]
    for cell in cells:
        index = 0
        for output in cells_json[cell].get("outputs", []):
            png = output.get("data", {}).get("image/png")
            if png is not None:
                yield cell, index, base64.b64decode(png)
                index += 1


def main():
    out = os.path.join(HERE, "notebook_figures")
    os.makedirs(out, exist_ok=True)
    count = 0
    for notebook, stem, cells in (("render.ipynb", "render", [3]), ("environment.ipynb", "environment", [6, 9, 12, 13, 14])):
        for cell, index, png in figures_of(notebook, cells):
            with open(os.path.join(out, f"{stem}_cell{cell}_{index}.png"), "wb") as f:
                f.write(png)
            count += 1
    print(count, "figures ->", out)


if __name__ == "__main__":
    main()
