"""Host-side file output off the training loop's critical path.

The reference's cadence block (bbhMahoGANy.py:1302-1382) pickles posterior samples and waveforms and saves three weight files inline
every `cadence` iterations; at its own defaults (batch 8, cadence 100) that is a noticeable share of the loop.  Here the device -> host
copies stay where they are (stream-ordered, cheap) and only the serialisation + file write is handed to ONE worker thread:

    with BackgroundWriter() as bg:
        ...
        bg.pickle(pe_samples, 'gan_pe_samples.sav', protocol=2)
        model.save_weights('generator.h5', True, writer=bg)

Jobs run in submission order (two writes of the same path end with the later content); an exception raised by a job is re-raised in the
caller's thread by the next submit() / flush() / close(), so a full disk does not pass silently.  Every file is written under
path + '.tmp' and moved into place with os.replace, so an interrupted run leaves either the previous or the new content under the
final name, never a truncated file.  Use it as a context manager (or close() in a finally block): the worker is a daemon thread, and only
close() guarantees that the queued jobs are on disk before the interpreter exits.
"""
import os
import pickle
import queue
import threading


class BackgroundWriter(object):
    def __init__(self, max_pending=8):
        self._q = queue.Queue(maxsize=max_pending)      # bounded: a slow disk throttles the loop instead of piling host copies up
        self._err = None
        self._closed = False
        self._t = threading.Thread(target=self._run, name='gennet-writer', daemon=True)
        self._t.start()

    def _run(self):
        while True:
            job = self._q.get()
            try:
                if job is None:
                    return
                if self._err is None:                   # after a failure the remaining jobs are dropped, the error is reported once
                    fn, args, kw = job
                    fn(*args, **kw)
            except BaseException as e:                  # noqa: B902 -- handed to the submitting thread
                self._err = e
            finally:
                self._q.task_done()

    def _check(self):
        if self._err is not None:
            e, self._err = self._err, None
            raise e

    def submit(self, fn, *args, **kw):
        if self._closed:
            raise RuntimeError('BackgroundWriter is closed')
        self._check()
        self._q.put((fn, args, kw))

    def pickle(self, obj, path, protocol=2):
        """pickle.dump(obj, open(path, 'wb'), protocol) in the worker; obj must not be mutated by the caller afterwards."""
        def job():
            tmp = path + '.tmp'
            try:
                with open(tmp, 'wb') as f:
                    pickle.dump(obj, f, protocol=protocol)
                os.replace(tmp, path)
            except BaseException:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise
        self.submit(job)

    def flush(self):
        self._q.join()
        self._check()

    def close(self):
        if not self._closed:
            self._closed = True
            self._q.put(None)
            self._t.join()
        self._check()

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is None:
            self.close()
        else:                                           # do not mask the caller's exception with a writer error
            try:
                self.close()
            except BaseException:                       # noqa: B902
                pass
        return False
