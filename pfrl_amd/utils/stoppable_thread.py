"""A ``threading.Thread`` that carries the event used to ask it to stop
(reference pfrl/utils/stoppable_thread.py).

The thread's ``target`` is expected to poll ``is_stopped()`` (or block in ``wait_stopped``) and
return on its own; nothing is interrupted from outside.  Several threads may share one event,
which is how a poller and a learner are stopped together.
"""
import threading


class StoppableThread(threading.Thread):
    def __init__(self, stop_event, *args, **kwargs):
        threading.Thread.__init__(self, *args, **kwargs)
        assert hasattr(stop_event, "set") and hasattr(stop_event, "is_set")
        self.stop_event = stop_event

    def is_stopped(self):
        """Has a stop been requested (by ``stop()`` here or by anyone holding the event)?"""
        return self.stop_event.is_set()

    def stop(self):
        """Request the stop; returns immediately, ``join()`` to wait for the thread."""
        self.stop_event.set()

    def wait_stopped(self, timeout=None):
        """Block until a stop is requested or ``timeout`` seconds pass; True iff requested."""
        return self.stop_event.wait(timeout)
