"""A single-machine stand-in for the two Toil objects the blast job functions touch
(toil is not installable here).  Only the members used by run_lastz /
make_chunked_alignments / combine_chunks exist: job.fileStore.{getLocalTempDir,getLocalTempFile,
readGlobalFile,writeGlobalFile,deleteGlobalFile,logToMaster}, job.cores, job.memory,
job.addChildJobFn / addFollowOnJobFn (run eagerly; .rv() returns the value)."""
from __future__ import annotations

import os
import shutil
import tempfile
import uuid


class FileID(str):
    """Toil FileIDs are strings with a .size (used for resource estimates, local_alignment.py:399-404)."""
    size = 0

    @staticmethod
    def of(path):
        f = FileID(path)
        f.size = os.path.getsize(path)
        return f


class _Promise:
    def __init__(self, value):
        self._v = value

    def rv(self):
        return self._v


class LocalFileStore:
    def __init__(self, root=None):
        self.root = root or tempfile.mkdtemp(prefix="miblast_jobstore_")
        self.messages = []

    def getLocalTempDir(self):
        return tempfile.mkdtemp(dir=self.root)

    def getLocalTempFile(self):
        fd, p = tempfile.mkstemp(dir=self.root)
        os.close(fd)
        return p

    def writeGlobalFile(self, path, cleanup=False):
        dst = os.path.join(self.root, "g_" + uuid.uuid4().hex)
        shutil.copyfile(path, dst)
        return FileID.of(dst)

    def readGlobalFile(self, file_id, userPath=None, mutable=False):
        if userPath is None:
            return str(file_id)
        shutil.copyfile(str(file_id), userPath)
        return userPath

    def deleteGlobalFile(self, file_id):
        try:
            os.remove(str(file_id))
        except OSError:
            pass

    def logToMaster(self, msg):
        self.messages.append(msg)


class LocalJob:
    def __init__(self, fileStore=None, cores=1, memory=2_000_000_000):
        self.fileStore = fileStore or LocalFileStore()
        self.cores = cores
        self.memory = memory
        self.children = []                     # what was asked for every child / follow-on job: {"fn", "cores", "memory", "disk", "accelerators"}

    def addChildJobFn(self, fn, *args, cores=None, memory=None, disk=None, accelerators=None, **kw):
        self.__dict__.setdefault("children", []).append({"fn": fn, "cores": cores, "memory": memory, "disk": disk, "accelerators": accelerators})
        child = LocalJob(self.fileStore, cores or self.cores, memory or self.memory)
        return _Promise(fn(child, *args, **kw))

    addFollowOnJobFn = addChildJobFn

    def addChild(self, other):
        """the reference hangs sub-graphs off a bare Job() (local_alignment.py:432-433): it works in its parent's file store"""
        other.fileStore, other.cores, other.memory = self.fileStore, self.cores, self.memory
        return other
