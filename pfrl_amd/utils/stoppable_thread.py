"""Thread that carries the event used to stop it (reference pfrl/utils/stoppable_thread.py)."""
import threading


class StoppableThread(threading.Thread):
    def __init__(self, stop_event, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.stop_event = stop_event

    def stop(self):
        self.stop_event.set()

    def is_stopped(self):
        return self.stop_event.is_set()
