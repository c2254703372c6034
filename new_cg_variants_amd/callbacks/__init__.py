"""History recorders, by the reference's names (numerical_experiments/callbacks/).

In the reference these are Python functions called with ``**locals()`` after every
iteration; each costs one or two extra host SpMVs.  Here they are *tags*: the solver
recognises them (by ``__name__``, so the reference's own callback functions work too)
and computes the quantity on the device inside the iteration loop.  Calling one directly
is an error -- there is no host implementation in the product.
"""


def _tag(name, doc):
    def recorder(**kwargs):
        raise RuntimeError(f'{name} is a device-side recorder tag: pass it in callbacks=[...] '
                           'of a new_cg_variants_amd solver')
    recorder.__name__ = name
    recorder.__doc__ = doc
    recorder.prcg_recorder = name
    return recorder


updated_residual_2_norm = _tag('updated_residual_2_norm',
                               '||r_k||_2 of the recurrence residual (callbacks/updated_residual_2_norm.py:40)')
residual_2_norm = _tag('residual_2_norm', '||b - A x_k||_2 (callbacks/residual_2_norm.py:41)')
error_A_norm = _tag('error_A_norm', 'sqrt(e^T A e), e = x_k - x_true (callbacks/error_A_norm.py:47-48)')
error_2_norm = _tag('error_2_norm', '||x_k - x_true||_2 (callbacks/error_2_norm.py:47-48)')


def print_k(K):
    """Progress printer, as callbacks/print_k.py:8-31: prints every K-th iteration."""
    def pk(**kwargs):
        k = kwargs['k']
        if k % K == 0:
            print(f"{kwargs['output']['name']}: iteration {k} of {kwargs['max_iter']}", end='\r')
    pk.prcg_host_light = True      # needs no vectors: never forces a device->host copy
    return pk


RECORDER_NAMES = ('updated_residual_2_norm', 'residual_2_norm', 'error_A_norm', 'error_2_norm')
