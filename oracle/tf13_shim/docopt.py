"""TEST INFRASTRUCTURE: the reference's model files do `from docopt import docopt` at import time (sparse:18);
docopt is not installed here and the CLI is out of scope, so importing must work and calling must not."""


def docopt(doc, argv=None, **kw):
    raise RuntimeError("docopt shim: the reference CLI is out of scope; construct the model with an args dict")
