class QiskitError(Exception):
    pass
