"""The one multiprocessing context the framework uses.  Every shared Value/Lock
must come from the *spawn* context: a semaphore created under the default fork
context is anonymous, and handing it to a spawned child rebuilds it from a raw
pointer (segfault).  The reference sidesteps this by calling
mp.set_start_method("spawn") before anything else (main.py:13); importing this
module makes library use (tests, bench) safe without touching global state.
"""
import torch.multiprocessing as mp

ctx = mp.get_context("spawn")
Value = ctx.Value
Lock = ctx.Lock
Process = ctx.Process
RawArray = ctx.RawArray
RawValue = ctx.RawValue
