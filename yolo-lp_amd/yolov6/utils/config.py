"""``Config.fromfile('configs/x.py')`` -> attribute-style nested dict.

Same surface as the reference's mmcv-style loader (yolov6/utils/config.py:33-101)
without the ``addict`` dependency (not installed in this image): a config file
is a python module whose top-level names become keys.
"""
import runpy


class ConfigDict(dict):
    """dict with attribute access; nested dicts are wrapped on the way in."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            return cls(v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError("'%s' object has no attribute '%s'" % (type(self).__name__, name))

    def __setattr__(self, name, value):
        self[name] = value


class Config(object):
    def __init__(self, cfg_dict=None, cfg_text=None, filename=None):
        if cfg_dict is None:
            cfg_dict = {}
        elif not isinstance(cfg_dict, dict):
            raise TypeError('cfg_dict must be a dict, but got {}'.format(type(cfg_dict)))
        object.__setattr__(self, '_cfg_dict', ConfigDict(cfg_dict))
        object.__setattr__(self, '_filename', filename)
        if not cfg_text and filename:
            with open(filename, 'r') as f:
                cfg_text = f.read()
        object.__setattr__(self, '_text', cfg_text or '')

    @staticmethod
    def fromfile(filename):
        filename = str(filename)
        if not filename.endswith('.py'):
            raise IOError('Only .py type are supported now!')
        names = {k: v for k, v in runpy.run_path(filename).items() if not k.startswith('__')}
        with open(filename, 'r') as f:
            text = filename + '\n' + f.read()
        return Config(names, cfg_text=text, filename=filename)

    filename = property(lambda self: self._filename)
    text = property(lambda self: self._text)

    def __repr__(self):
        return 'Config (path: {}): {}'.format(self.filename, dict.__repr__(self._cfg_dict))

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __setattr__(self, name, value):
        self._cfg_dict[name] = value

    def __contains__(self, name):
        return name in self._cfg_dict
