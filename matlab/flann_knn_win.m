function [idx, dist] = flann_knn_win(train, query, k, method, varargin) %#ok<INUSD>
    %FLANN_KNN_WIN Replaces the reference's Windows-only mex (PP/mex/flann_knn.cpp) on the device: exact kNN.
    %   single descriptors: squared-L2 (in place of the randomized kd-forest, :226-233); uint8 descriptors: Hamming
    %   (in place of BFMatcher knnMatch 'bf', :199-223, and of the LSH index 'flann', :235-240).  method / trees / checks
    %   are accepted and ignored except for the type check the reference makes for 'bf'.
    if nargin >= 4 && strcmp(char(method), 'bf') && ~isa(train, 'uint8')
        error('flann_knn:bf', 'BFMatcher only supports uint8 (binary) descriptors');
    end
    if isa(train, 'uint8')
        [idx, dist] = aps_mex('knn_global', train, uint8(query), double(k));
    else
        [idx, dist] = aps_mex('knn_global', single(train), single(query), double(k));
    end
end
