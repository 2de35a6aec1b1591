function varargout = aps_call_shadowed(name, shadowFile, varargin)
    %APS_CALL_SHADOWED Calls the implementation of NAME that this overlay shadows (the reference's own file).
    %   shadowFile = mfilename('fullpath') of the calling shadow.  The first NAME on the MATLAB path that does not live
    %   in the overlay folder is the reference's; its folder is made current for the duration of the call (the current
    %   folder precedes every path entry, so feval(NAME) resolves to that file) and restored afterwards, also on error.
    here = fileparts(shadowFile);
    cands = which(name, '-all');
    target = '';
    for k = 1:numel(cands)
        [folder, ~, ext] = fileparts(cands{k});
        if ~strcmp(folder, here) && any(strcmp(ext, {'.m', '.p', '.mlx'}))
            target = folder;
            break;
        end
    end
    if isempty(target)
        error('aps:shadow', '%s: the reference implementation is not on the MATLAB path below the overlay folder.', name);
    end
    old = cd(target);
    restore = onCleanup(@() cd(old)); %#ok<NASGU>
    [varargout{1:nargout}] = feval(name, varargin{:});
end
