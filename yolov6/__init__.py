"""Drop-in import root: ``import yolov6.*`` from the repo root resolves to the
host-side mirror kept in ``yolo-lp_amd/yolov6`` (the reference's tools do
``sys.path.append(os.getcwd())`` and import ``yolov6.*``, tools/infer.py:10-15)."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                 'yolo-lp_amd', 'yolov6'))
